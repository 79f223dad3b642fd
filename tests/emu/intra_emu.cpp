// tests/emu/intra_emu.cpp -- TEST INFRASTRUCTURE.  Compiles the product's e264_intra_kernel (edge264_amd/csrc/e264_intra.h) for the
// host AS IT IS and runs it: the 64 lanes of a wave are 64 fibres (ucontext) that run freely between the collectives and meet
// at them -- wave_sync() is a barrier, E264_BALLOT gathers one bit per lane, E264_FIRST hands out lane 0's value -- which is
// all the source assumes about a wave (lanes communicate through LDS across wave_sync only).  The picture's workgroup is ONE
// wave (the kernel's NW = 1 instantiation) taking the macroblock rows in order, so a row never waits for the row above.
// tests/test_intra_emu.py compares whole pictures with the CPU oracle: a logic error in the intra kernel is found here, on
// the host, and not on the GPU box.
#include "emu_shims.h"
#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>

// ---- the wave: 64 fibres, round-robin, switching only inside the collectives -----------------------------------------
namespace {
enum { LANES = 64, STACK = 512 * 1024 };
ucontext_t g_main, g_ctx[LANES];
char *g_stack[LANES];
bool g_done[LANES];
int g_cur, g_alive, g_arrived;
unsigned g_gen;

void switch_to_next()
{ // the next fibre that has not finished (this one, if it is the only one left, just goes on)
	const int from = g_cur;
	for (int k = 1; k <= LANES; k++) {
		const int n = (from + k) % LANES;
		if (g_done[n]) continue;
		if (n == from) return;
		g_cur = n;
		swapcontext(&g_ctx[from], &g_ctx[n]);
		return;
	}
}
} // namespace

static inline void wave_sync()
{
	const unsigned gen = g_gen;
	if (++g_arrived == g_alive) { g_arrived = 0; g_gen++; return; } // the last one to arrive releases everybody
	while (g_gen == gen) switch_to_next();
}
static unsigned g_first_cnt[LANES], g_bal_cnt[LANES];
static uint32_t g_first_val[3];
static unsigned long long g_bal[3];
static inline uint32_t emu_first(uint32_t v)
{ // v_readfirstlane: every lane is active wherever the kernel uses it, so "first" is lane 0.  Rotating slots: a lane may be one call ahead
	const unsigned n = g_first_cnt[g_cur]++ % 3;
	if (g_cur == 0) g_first_val[n] = v;
	wave_sync();
	return g_first_val[n];
}
static inline unsigned long long emu_ballot(bool p)
{
	const unsigned n = g_bal_cnt[g_cur]++ % 3;
	if (g_cur == 0) g_bal[(n + 1) % 3] = 0; // the slot of the NEXT ballot: nobody is there yet, everybody has left its previous use
	if (p) g_bal[n] |= 1ull << g_cur;
	wave_sync();
	return g_bal[n];
}
#define E264_FIRST(x) ((__typeof__((x) + 0))emu_first((uint32_t)(x)))
#define E264_BALLOT(x) emu_ballot(x)
#define E264_WG_SYNC() wave_sync()                 /* the workgroup is one wave */
#define E264_SLEEP() (fprintf(stderr, "intra_emu: a row waits for the row above: impossible with one wave taking the rows in order\n"), abort())
#define E264_FENCE_ACQUIRE() do { } while (0)
#define E264_FENCE_RELEASE() do { } while (0)
#define E264_PROGRESS_STORE(p, v) (*(p) = (v))
#define E264_PROGRESS_LOAD(p) (*(p))
#define E264_ROW_TAKE(p) ((*(p))++)                 /* (never reached: one wave takes the rows in order, intra_next_row's NW == 1 path) */
#define PH_DECL
#define PH(k)
#define PH_PARAMS
#define PH_ARGS
static inline uint32_t v_sad_u8(uint32_t a, uint32_t b, uint32_t c)
{
	for (int i = 0; i < 4; i++) { const int d = (int)(a >> (8 * i) & 255) - (int)(b >> (8 * i) & 255); c += (uint32_t)(d < 0 ? -d : d); }
	return c;
}
static inline int relane(int lane) { return lane; }

#include "../../edge264_amd/csrc/e264_intra.h"

namespace {
IntraLds<1> g_lds;
static uint8_t *g_expand; // expansion buffer of a wire packet (include/edge264_compact.h), filled by the caller (pred_emu's e264emu_expand); NULL for version 4
extern "C" __attribute__((visibility("default"))) void e264emu_set_expand(uint8_t *area) { g_expand = area; }
const E264Job *g_job;
int g_planes = 3; // 3: e264_intra_kernel; 1 / 2: one workgroup of e264_intra_planes_kernel (luma / chroma)
void fibre_main(int lane)
{
	if (g_planes == 1) intra_kernel_body<1, 1>(g_lds, *g_job, lane);
	else if (g_planes == 2) intra_kernel_body<1, 2>(g_lds, *g_job, lane);
	else intra_kernel_body<1>(g_lds, *g_job, lane);
	g_done[lane] = true;
	g_alive--;
	if (g_alive > 0 && g_arrived == g_alive) { g_arrived = 0; g_gen++; } // (never the case: the lanes of a wave leave the kernel together)
	for (int k = 1; k < LANES; k++) {
		const int n = (lane + k) % LANES;
		if (!g_done[n]) { g_cur = n; setcontext(&g_ctx[n]); }
	}
	setcontext(&g_main);
}
} // namespace

// e264_intra_kernel<1> on one picture: every intra macroblock of the packet is reconstructed into dpb[dst_slot]
// scratch: NULL (every chunk of every row is scanned), or the stream's scratch with the intra bitmap e264_pred_kernel has left there
// (tests/emu/pred_emu.cpp e264emu_pred_frame2 on the same packet): rows and chunks without a bit are left alone
extern "C" __attribute__((visibility("default"))) int e264emu_intra_frame2(const uint8_t *pkt, uint8_t *const *dpb, uint8_t *scratch);
extern "C" __attribute__((visibility("default"))) int e264emu_intra_frame(const uint8_t *pkt, uint8_t *const *dpb)
{
	return e264emu_intra_frame2(pkt, dpb, nullptr);
}
extern "C" __attribute__((visibility("default"))) int e264emu_intra_frame2(const uint8_t *pkt, uint8_t *const *dpb, uint8_t *scratch)
{
	const E264Job job = {pkt, dpb, scratch, g_expand};
	FrameCtx f;
	if (!open_frame(f, job))
		return -1;
	g_job = &job;
	memset(&g_lds, 0xA5, sizeof(g_lds)); // LDS is not zeroed on the device either
	memset(g_first_cnt, 0, sizeof(g_first_cnt)); memset(g_bal_cnt, 0, sizeof(g_bal_cnt)); memset(g_bal, 0, sizeof(g_bal));
	g_alive = LANES; g_arrived = 0; g_gen = 0;
	for (int lane = 0; lane < LANES; lane++) {
		if (!g_stack[lane]) g_stack[lane] = (char *)malloc(STACK);
		g_done[lane] = false;
		getcontext(&g_ctx[lane]);
		g_ctx[lane].uc_stack.ss_sp = g_stack[lane];
		g_ctx[lane].uc_stack.ss_size = STACK;
		g_ctx[lane].uc_link = &g_main;
		makecontext(&g_ctx[lane], (void (*)())fibre_main, 1, lane);
	}
	g_cur = 0;
	swapcontext(&g_main, &g_ctx[0]);
	return 0;
}

// e264_intra_planes_kernel on one picture: its two workgroups one after the other -- chroma FIRST, so that a chroma sample that needed a luma one (none may) would find it missing
extern "C" __attribute__((visibility("default"))) int e264emu_intra_frame_planes(const uint8_t *pkt, uint8_t *const *dpb)
{
	g_planes = 2;
	int r = e264emu_intra_frame2(pkt, dpb, nullptr);
	g_planes = 1;
	if (!r) r = e264emu_intra_frame2(pkt, dpb, nullptr);
	g_planes = 3;
	return r;
}
