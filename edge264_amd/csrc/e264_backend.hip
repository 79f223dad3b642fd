// e264_backend.hip -- C-ABI of the MI355X reconstruction back end (include/edge264_hip.h).
//
// Host side of the drop-in boundary: owns the device DPB of every decoder ("stream"), moves
// command packets to HBM with pinned async copies, launches the frame kernels on one HIP
// queue per device, and copies finished frames back on demand.  No torch types, no CPU
// fallback: if there is no gfx950 device every entry point fails with ENODEV.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <thread>
#include <string>
#include <vector>

#include "../../include/edge264_hip.h"
#include "e264_kernels.h"

#define API extern "C" __attribute__((visibility("default")))

static thread_local char g_err[256];
static int fail(int code, const char *what, hipError_t e = hipSuccess)
{
	snprintf(g_err, sizeof(g_err), "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
	return code;
}
#define HIPCHK(call, code) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(code, #call, e_); } while (0)

API const char *e264hip_last_error(void) { return g_err; }
// events a host thread waits on: blocking, so that the waiter sleeps instead of spinning on a core the parser threads could use
#define E264_WAIT_EVENT (hipEventDisableTiming | hipEventBlockingSync)

// A few host threads for the per-packet work of a batch that arrives in ordinary host memory (validation of every
// macroblock record + the copy into page-locked staging memory: 0.18 ms per 1080p packet on one thread = 5 k frames/s,
// while PCIe carries 28 k).  No HIP call is ever made from these threads.  E264_HOST_THREADS overrides the count (0: none).
namespace {
struct HostPool {
	std::vector<std::thread> th;
	std::mutex m;
	std::condition_variable cv, done_cv;
	const std::function<void(int)> *fn = nullptr;
	std::atomic<int> next{0};
	int n = 0, active = 0;
	uint64_t gen = 0;
	bool stop = false, started = false;
	void run() { for (int i; (i = next.fetch_add(1)) < n;) (*fn)(i); }
	void worker()
	{
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lk(m);
		for (;;) {
			cv.wait(lk, [&] { return stop || gen != seen; });
			if (stop) return;
			seen = gen;
			lk.unlock();
			run();
			lk.lock();
			if (--active == 0) done_cv.notify_one();
		}
	}
	void parallel_for(int count, const std::function<void(int)> &f)
	{
		std::unique_lock<std::mutex> lk(m);
		if (!started) {
			started = true;
			const char *e = getenv("E264_HOST_THREADS");
			int want = e ? atoi(e) : (int)std::min(15u, std::thread::hardware_concurrency() / 2);
			for (int i = 0; i < want; i++) th.emplace_back([this] { worker(); });
		}
		if (th.empty() || count < 4) { lk.unlock(); for (int i = 0; i < count; i++) f(i); return; }
		fn = &f; n = count; next = 0; active = (int)th.size(); gen++;
		lk.unlock();
		cv.notify_all();
		run(); // the caller works too
		lk.lock();
		done_cv.wait(lk, [&] { return active == 0; });
	}
	~HostPool()
	{
		{ std::lock_guard<std::mutex> lk(m); stop = true; }
		cv.notify_all();
		for (auto &t : th) t.join();
	}
};
HostPool g_pool;
std::mutex g_pool_user; // one batch at a time uses the pool
}

struct E264Packet {
	E264Device *dev;
	uint8_t *d_bytes;
	size_t bytes;
	int dst_slot;
	int n_mbs, n_tiles;
	uint64_t frame_bytes;  // plane_size_Y + plane_size_C the kernels will touch in every slot the packet names
	uint32_t ref_mask;     // DPB slots its motion refers to
};

struct E264Device {
	int ordinal;
	hipStream_t q;
	int waves;                 // waves per frame workgroup of the deblocking kernel (5 macroblock rows each): 2, 4, 7 or 8
	int intra_waves;           // waves per frame workgroup of the intra kernel (1 macroblock row each)
	int dbg_mode;
	int side_queue;            // option "side_queue": parameter kernel on a second queue beside the macroblock-parallel kernel
	hipStream_t q2;
	hipEvent_t forked, joined;
	std::mutex lock;           // kernel launches + their timing marks
	std::mutex batch_lock;     // e264hip_submit_batch_host: job ring
	hipEvent_t ev[16];
	// per-launch kernel timing
	bool ktiming;
	struct Marks { hipEvent_t e[5], a[2]; bool side; };
	std::vector<Marks> kev;
	size_t kev_used;
	// job tables of host-packet batches (e264hip_submit_batch_host): a ring of pinned + device buffers
	struct JobRing { E264Job *h = nullptr, *d = nullptr; int cap = 0; hipEvent_t done = nullptr; bool busy = false; } jring[4];
	int jring_next = 0;
	// Which submission wrote a slot last, and when it has retired: edge264_get_frame of ONE decoder must not wait for the
	// whole device (every later batch of every other decoder), only for the submission that produced its frame.
	hipStream_t qc = nullptr;  // copy queue of the downloads (D2H beside the kernels)
	uint64_t serial = 0;       // submissions so far (guarded by `lock`)
	enum { NEV = 64 };
	hipEvent_t sub_ev[NEV] = {};
	uint64_t sub_serial[NEV] = {};
};

struct E264Stream {
	E264Device *dev;
	uint8_t **d_table;                    // device array [E264_MAX_SLOTS] of slot pointers
	uint8_t *h_table[E264_MAX_SLOTS];     // same, host copy
	void *mirror[E264_MAX_SLOTS];         // pinned host mirrors
	size_t slot_bytes[E264_MAX_SLOTS];
	uint8_t *d_dbk;                       // deblocking parameters, E264_DBK_BYTES per macroblock
	size_t dbk_mbs;
	// packet staging ring (pinned host) + device copies
	struct Stage { void *h; uint8_t *d; size_t cap; hipEvent_t done; bool busy; E264Job *d_job; } stage[4];
	int stage_next;
	uint64_t slot_serial[E264_MAX_SLOTS]; // submission that wrote the slot last (0: none since it was allocated)
	hipEvent_t dl_done;                   // the stream's last download
};

static int set_device(E264Device *dev)
{
	HIPCHK(hipSetDevice(dev->ordinal), EIO);
	return 0;
}

API int e264hip_device_open(int ordinal, E264Device **out)
{
	if (!out) return fail(EINVAL, "null out");
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(ENODEV, "no HIP device");
	if (ordinal < 0 || ordinal >= n) return fail(ENODEV, "device ordinal out of range");
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, ordinal), EIO);
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		snprintf(g_err, sizeof(g_err), "device %d is %s, this back end is built for gfx950 only", ordinal, prop.gcnArchName);
		return ENODEV;
	}
	E264Device *d = new (std::nothrow) E264Device();
	if (!d) return fail(ENOMEM, "device object");
	d->ordinal = ordinal;
	d->waves = 7; // 35 macroblock rows in flight: a 1080p picture in two even rounds
	d->intra_waves = 16; // 16 rows in flight: 1.6 -> 1.1 ms per 256-frame launch (the intra kernel fits 128 VGPRs)
	d->dbg_mode = 0;
	d->ktiming = false; d->kev_used = 0;
	if (hipSetDevice(ordinal) != hipSuccess || hipStreamCreateWithFlags(&d->q, hipStreamNonBlocking) != hipSuccess) {
		delete d;
		return fail(EIO, "hipStreamCreate");
	}
	for (int i = 0; i < 16; i++)
		hipEventCreate(&d->ev[i]);
	d->side_queue = 0; d->q2 = nullptr; d->forked = d->joined = nullptr; // measured: no gain (profiles/r01k_ablation_breakdown.txt), off by default
	if (hipStreamCreateWithFlags(&d->q2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d->forked, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->joined, hipEventDisableTiming) != hipSuccess) {
		d->q2 = nullptr; // not fatal: the option then stays off
	}
	if (hipStreamCreateWithFlags(&d->qc, hipStreamNonBlocking) != hipSuccess) d->qc = nullptr; // downloads then share the kernels' queue
	for (int i = 0; i < E264Device::NEV; i++)
		if (hipEventCreateWithFlags(&d->sub_ev[i], E264_WAIT_EVENT) != hipSuccess) d->sub_ev[i] = nullptr;
	*out = d;
	return 0;
}

API void e264hip_device_close(E264Device *dev)
{
	if (!dev) return;
	hipSetDevice(dev->ordinal);
	hipStreamSynchronize(dev->q);
	for (int i = 0; i < 16; i++) hipEventDestroy(dev->ev[i]);
	for (auto &p : dev->kev) { for (int i = 0; i < 5; i++) hipEventDestroy(p.e[i]); for (int i = 0; i < 2; i++) hipEventDestroy(p.a[i]); }
	if (dev->q2) { hipStreamSynchronize(dev->q2); hipStreamDestroy(dev->q2); }
	if (dev->forked) hipEventDestroy(dev->forked);
	if (dev->joined) hipEventDestroy(dev->joined);
	for (auto &jr : dev->jring) {
		if (jr.h) hipHostFree(jr.h);
		if (jr.d) hipFree(jr.d);
		if (jr.done) hipEventDestroy(jr.done);
	}
	if (dev->qc) { hipStreamSynchronize(dev->qc); hipStreamDestroy(dev->qc); }
	for (int i = 0; i < E264Device::NEV; i++) if (dev->sub_ev[i]) hipEventDestroy(dev->sub_ev[i]);
	hipStreamDestroy(dev->q);
	delete dev;
}

API int e264hip_device_sync(E264Device *dev)
{
	if (!dev) return fail(EINVAL, "null device");
	if (set_device(dev)) return EIO;
	HIPCHK(hipStreamSynchronize(dev->q), EIO);
	if (dev->q2) HIPCHK(hipStreamSynchronize(dev->q2), EIO);
	return 0;
}

API int e264hip_set_option(E264Device *dev, const char *name, int value)
{
	if (!dev || !name) return -1;
	if (!strcmp(name, "debug_mode")) { // profiling ablation bits OR-ed into the kernels' mode argument
		int prev = dev->dbg_mode;
		dev->dbg_mode = value;
		return prev;
	}
	if (!strcmp(name, "side_queue")) {
		int prev = dev->side_queue;
		dev->side_queue = value && dev->q2;
		return prev;
	}
	if (!strcmp(name, "intra_waves")) {
		int prev = dev->intra_waves;
		if (value == 4 || value == 8 || value == 16) dev->intra_waves = value;
		return prev;
	}
	if (!strcmp(name, "waves")) {
		int prev = dev->waves;
		if (value == 2 || value == 4 || value == 7 || value == 8) dev->waves = value; // anything else keeps the setting
		return prev;
	}
	return -1;
}

API int e264hip_stream_open(E264Device *dev, E264Stream **out)
{
	if (!dev || !out) return fail(EINVAL, "null argument");
	if (set_device(dev)) return EIO;
	E264Stream *s = new (std::nothrow) E264Stream();
	if (!s) return fail(ENOMEM, "stream object");
	memset(s, 0, sizeof(*s));
	s->dev = dev;
	if (hipMalloc(&s->d_table, sizeof(uint8_t *) * E264_MAX_SLOTS) != hipSuccess) { delete s; return fail(ENOMEM, "slot table"); }
	hipMemsetAsync(s->d_table, 0, sizeof(uint8_t *) * E264_MAX_SLOTS, dev->q);
	*out = s;
	return 0;
}

API int e264hip_stream_flush(E264Stream *s)
{
	if (!s) return fail(EINVAL, "null stream");
	return e264hip_device_sync(s->dev);
}

API void e264hip_stream_close(E264Stream *s)
{
	if (!s) return;
	e264hip_device_sync(s->dev);
	for (int i = 0; i < E264_MAX_SLOTS; i++) {
		if (s->h_table[i]) hipFree(s->h_table[i]);
		if (s->mirror[i]) hipHostFree(s->mirror[i]);
	}
	for (auto &st : s->stage) {
		if (st.h) hipHostFree(st.h);
		if (st.d) hipFree(st.d);
		if (st.d_job) hipFree(st.d_job);
		if (st.done) hipEventDestroy(st.done);
	}
	if (s->d_dbk) hipFree(s->d_dbk);
	if (s->dl_done) hipEventDestroy(s->dl_done);
	hipFree(s->d_table);
	delete s;
}

static int push_table(E264Stream *s)
{
	// small synchronous-looking update, ordered on the queue before any kernel that reads it
	HIPCHK(hipMemcpyAsync(s->d_table, s->h_table, sizeof(uint8_t *) * E264_MAX_SLOTS, hipMemcpyHostToDevice, s->dev->q), EIO);
	HIPCHK(hipStreamSynchronize(s->dev->q), EIO); // h_table is pageable and may change right after
	return 0;
}

API int e264hip_frame_alloc(E264Stream *s, int slot, size_t samples_bytes, void **host_mirror)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS || samples_bytes == 0) return fail(EINVAL, "frame_alloc arguments");
	if (set_device(s->dev)) return EIO;
	if (s->h_table[slot]) e264hip_frame_free(s, slot);
	// +64: the MC fast path reads whole dwords up to 3 bytes past a 9-sample span (like the
	// reference's +16 over-read margin, src/edge264_headers.c:115)
	if (hipMalloc(&s->h_table[slot], samples_bytes + 64) != hipSuccess) { s->h_table[slot] = nullptr; return fail(ENOMEM, "hipMalloc frame"); }
	s->slot_bytes[slot] = samples_bytes;
	s->slot_serial[slot] = 0;
	if (host_mirror) {
		if (hipHostMalloc(&s->mirror[slot], samples_bytes, hipHostMallocDefault) != hipSuccess) {
			hipFree(s->h_table[slot]); s->h_table[slot] = nullptr; s->mirror[slot] = nullptr;
			return fail(ENOMEM, "hipHostMalloc mirror");
		}
		*host_mirror = s->mirror[slot];
	}
	return push_table(s);
}

API void e264hip_frame_free(E264Stream *s, int slot)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS || !s->h_table[slot]) return;
	set_device(s->dev);
	hipStreamSynchronize(s->dev->q);
	hipFree(s->h_table[slot]);
	if (s->mirror[slot]) hipHostFree(s->mirror[slot]);
	s->h_table[slot] = nullptr; s->mirror[slot] = nullptr; s->slot_bytes[slot] = 0;
	push_table(s);
}

API int e264hip_frame_fill(E264Stream *s, int slot, int value)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS || !s->h_table[slot]) return fail(EINVAL, "frame_fill slot");
	if (set_device(s->dev)) return EIO;
	HIPCHK(hipMemsetAsync(s->h_table[slot], value, s->slot_bytes[slot], s->dev->q), EIO);
	s->slot_serial[slot] = 0; // written outside a submission: a later wait falls back to the whole queue
	return 0;
}

API int e264hip_frame_upload(E264Stream *s, int slot, const void *src, size_t bytes)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS || !s->h_table[slot] || bytes > s->slot_bytes[slot]) return fail(EINVAL, "frame_upload");
	if (set_device(s->dev)) return EIO;
	HIPCHK(hipMemcpyAsync(s->h_table[slot], src, bytes, hipMemcpyHostToDevice, s->dev->q), EIO);
	HIPCHK(hipStreamSynchronize(s->dev->q), EIO);
	s->slot_serial[slot] = 0;
	return 0;
}

// tiles: workgroups e264_pred_kernel needs for this frame (16 x 8 macroblocks each)
static int check_packet(const void *packet, size_t bytes, int *dst, int *n_mbs, int *tiles = nullptr)
{
	const E264FrameHdr *h = (const E264FrameHdr *)packet;
	if (!packet || bytes < sizeof(*h) || h->magic != E264_MAGIC || h->version != E264_VERSION || h->total_bytes > bytes)
		return fail(EINVAL, "not a command packet");
	if (h->dst_slot < 0 || h->dst_slot >= E264_MAX_SLOTS) return fail(EINVAL, "dst_slot");
	size_t n_mb = (size_t)h->width_mbs * h->height_mbs;
	size_t need = (size_t)h->mbs_off + n_mb * sizeof(E264Mb);
	if (h->motion_off && (h->motion_off < need || (need = (size_t)h->motion_off) > h->total_bytes)) return fail(EINVAL, "motion section");
	if (need > h->payload_off || (size_t)h->payload_off + h->payload_bytes > h->total_bytes) return fail(EINVAL, "packet layout");
	*dst = h->dst_slot;
	*n_mbs = (int)h->width_mbs * h->height_mbs;
	if (tiles) *tiles = ((h->width_mbs + 15) / 16) * ((h->height_mbs + 7) / 8);
	return 0;
}

// Everything a kernel will dereference through the packet, checked on the host before the packet may reach the
// device (a wild offset would be a GPU memory fault = process abort, not an error code): section layout, per-macroblock
// kind / slice index / payload bounds, reference slots.  `slots` (may be null): allocated-slot table of the stream.
// `slot_bytes` (with `slots`): size of every allocated slot -- a packet whose header claims a larger picture than the slot
// it writes or reads (SPS size change, stale capture, foreign packet) would make the kernels run past the allocation.
// `ref_mask_out` (may be null): DPB slots the packet's motion refers to.
static int check_packet_deep(const void *packet, size_t bytes, uint8_t *const *slots, const size_t *slot_bytes = nullptr, uint32_t *ref_mask_out = nullptr)
{
	int dst, n_mbs, r = check_packet(packet, bytes, &dst, &n_mbs);
	if (r) return r;
	const E264FrameHdr *h = (const E264FrameHdr *)packet;
	const uint8_t *p = (const uint8_t *)packet;
	if (h->width_mbs == 0 || h->height_mbs == 0 || h->height_mbs > 1056) return fail(EINVAL, "frame size");
	if (h->n_slices == 0 || (size_t)h->slices_off + (size_t)h->n_slices * sizeof(E264SliceParams) > h->mbs_off) return fail(EINVAL, "slice section");
	if ((h->slices_off | h->mbs_off | h->motion_off | h->payload_off) & 7) return fail(EINVAL, "section alignment");
	if (h->stride_Y < (uint32_t)h->width_mbs * 16 || h->stride_C < (uint32_t)h->width_mbs * 16 || (h->stride_Y & 15) || (h->stride_C & 7))
		return fail(EINVAL, "strides");
	if ((uint64_t)h->plane_size_Y < (uint64_t)h->stride_Y * h->height_mbs * 16 || (uint64_t)h->plane_size_C < (uint64_t)h->stride_C * h->height_mbs * 8)
		return fail(EINVAL, "plane sizes");
	const uint64_t frame_need = (uint64_t)h->plane_size_Y + h->plane_size_C;
	if (slots && slot_bytes && slots[dst] && frame_need > slot_bytes[dst]) return fail(EINVAL, "picture larger than the destination slot");
	const E264Mb *mbs = (const E264Mb *)(p + h->mbs_off);
	const uint8_t *mot = h->motion_off ? p + h->motion_off : nullptr; // compact motion records, up to payload_off
	const uint32_t mot_bytes = h->motion_off ? h->payload_off - h->motion_off : 0;
	uint32_t ref_mask = 0;
	for (int a = 0; a < n_mbs; a++) {
		const E264Mb &m = mbs[a];
		if (m.kind > E264_MB_INTER) return fail(EINVAL, "macroblock kind");
		if (m.slice >= h->n_slices || m.dbk_slice >= h->n_slices) return fail(EINVAL, "macroblock slice index"); // every record: the parameter kernel reads the slice of absent macroblocks too
		if (m.kind == E264_MB_ABSENT) continue;
		if ((m.flags & E264_MBF_T8x8) && (m.kind == E264_MB_I16x16 || m.kind == E264_MB_PCM)) return fail(EINVAL, "8x8 transform flag on an Intra16x16 / PCM macroblock");
		if ((m.payload_off & 7) || (uint64_t)m.payload_off + e264_mb_payload_bytes(&m) > h->payload_bytes) return fail(EINVAL, "macroblock payload");
		if ((m.flags & E264_MBF_EDGE_LEFT) && a % h->width_mbs == 0) return fail(EINVAL, "left edge flag on the first column");
		if ((m.flags & E264_MBF_EDGE_TOP) && a < h->width_mbs) return fail(EINVAL, "top edge flag on the first row");
		if (m.kind == E264_MB_INTER) {
			if (!mot) return fail(EINVAL, "inter macroblock without motion section");
			uint32_t d[2];
			memcpy(d, m.modes, 8); // motion directory: record offset, shape
			if ((d[0] & 3) || d[1] >> 26 || (uint64_t)d[0] + e264_mot_record_bytes(d[1]) > mot_bytes) return fail(EINVAL, "macroblock motion record");
			E264Motion mx;
			e264_motion_expand(d[1], mot + d[0], &mx);
			for (int i = 0; i < 8; i++) {
				int rp = mx.refPic[i];
				const bool used = E264_MOT_UNI(d[1], i >> 2) || E264_MOT_USED(d[1], i);
				if (rp < (used ? 0 : -1) || rp >= E264_MAX_SLOTS) return fail(EINVAL, "reference slot"); // a part the directory announces predicts from a picture
				if (rp >= 0 && slots && !slots[rp]) return fail(EINVAL, "reference slot not allocated");
				if (rp >= 0 && slots && slot_bytes && frame_need > slot_bytes[rp]) return fail(EINVAL, "picture larger than a reference slot");
				if (rp >= 0) ref_mask |= 1u << rp;
				if (mx.refIdx[i] < -1 || mx.refIdx[i] > 31) return fail(EINVAL, "reference index");
			}
		}
	}
	if (ref_mask_out) *ref_mask_out = ref_mask;
	return 0;
}

// host-only entry point of the same checks (tests, front ends that want to vet a capture file)
API int e264hip_packet_check(const void *packet, size_t bytes)
{
	return check_packet_deep(packet, bytes, nullptr);
}

static int ensure_dbk(E264Stream *s, int n_mbs)
{
	if (s->dbk_mbs >= (size_t)n_mbs) return 0;
	if (s->d_dbk) { hipStreamSynchronize(s->dev->q); hipFree(s->d_dbk); s->d_dbk = nullptr; s->dbk_mbs = 0; }
	if (hipMalloc((void **)&s->d_dbk, (size_t)n_mbs * E264_DBK_BYTES) != hipSuccess) return fail(ENOMEM, "hipMalloc deblock parameters");
	s->dbk_mbs = (size_t)n_mbs;
	return 0;
}

// Launches the kernels over a job table that already lives in HBM.
static int launch(E264Device *dev, const E264Job *d_jobs, int n, int max_mbs, int max_tiles, int mode, uint64_t *serial_out = nullptr)
{
	std::lock_guard<std::mutex> g(dev->lock);
	hipEvent_t *marks = nullptr;
	E264Fork fork = {dev->side_queue ? dev->q2 : nullptr, dev->forked, dev->joined, nullptr};
	if (dev->ktiming) {
		if (dev->kev_used == dev->kev.size()) {
			E264Device::Marks m;
			for (int i = 0; i < 5; i++) hipEventCreate(&m.e[i]);
			for (int i = 0; i < 2; i++) hipEventCreate(&m.a[i]);
			dev->kev.push_back(m);
		}
		E264Device::Marks &m = dev->kev[dev->kev_used++];
		m.side = fork.aux != nullptr && (mode & 2) && !((mode | dev->dbg_mode) & 2048);
		marks = m.e; fork.amarks = m.a;
	}
	HIPCHK(e264_launch_frames(d_jobs, n, max_mbs, max_tiles, mode | dev->dbg_mode, dev->waves | dev->intra_waves << 8, dev->q, marks, &fork), EIO);
	const uint64_t serial = ++dev->serial;
	const int idx = (int)(serial % E264Device::NEV);
	if (dev->sub_ev[idx] && hipEventRecord(dev->sub_ev[idx], dev->q) == hipSuccess) dev->sub_serial[idx] = serial;
	else dev->sub_serial[idx] = 0;
	if (serial_out) *serial_out = serial;
	return 0;
}

// Blocks until the submission that wrote `slot` last has retired (not until the device is idle).
static int wait_slot(E264Stream *s, int slot)
{
	E264Device *dev = s->dev;
	const uint64_t serial = s->slot_serial[slot];
	hipEvent_t ev = nullptr;
	if (serial) {
		std::lock_guard<std::mutex> g(dev->lock);
		const int idx = (int)(serial % E264Device::NEV);
		if (dev->sub_serial[idx] == serial) ev = dev->sub_ev[idx];
	}
	if (!ev) return e264hip_device_sync(dev); // filled / uploaded outside a submission, or the event ring has wrapped: everything queued retires
	HIPCHK(hipEventSynchronize(ev), EIO);
	return 0;
}

API void *e264hip_packet_buffer(E264Stream *s, size_t max_bytes)
{
	if (!s || set_device(s->dev)) return nullptr;
	E264Stream::Stage &st = s->stage[s->stage_next];
	if (st.busy) { hipEventSynchronize(st.done); st.busy = false; }
	if (!st.done && hipEventCreateWithFlags(&st.done, E264_WAIT_EVENT) != hipSuccess) { st.done = nullptr; fail(EIO, "hipEventCreate"); return nullptr; }
	if (!st.d_job && hipMalloc((void **)&st.d_job, sizeof(E264Job)) != hipSuccess) { st.d_job = nullptr; fail(ENOMEM, "job slot"); return nullptr; }
	if (st.cap < max_bytes) {
		if (st.h) hipHostFree(st.h);
		if (st.d) hipFree(st.d);
		st.h = nullptr; st.d = nullptr; st.cap = 0;
		size_t cap = (max_bytes + 65535) & ~(size_t)65535;
		if (hipHostMalloc(&st.h, cap + 64, hipHostMallocDefault) != hipSuccess) { st.h = nullptr; fail(ENOMEM, "pinned packet buffer"); return nullptr; }
		if (hipMalloc((void **)&st.d, cap) != hipSuccess) { hipHostFree(st.h); st.h = nullptr; st.d = nullptr; fail(ENOMEM, "device packet buffer"); return nullptr; }
		st.cap = cap;
	}
	return st.h;
}

API int e264hip_frame_submit(E264Stream *s, const void *packet, size_t bytes)
{
	if (!s) return fail(EINVAL, "null stream");
	int dst, n_mbs, n_tiles, r = check_packet(packet, bytes, &dst, &n_mbs, &n_tiles);
	if (r) return r;
	if (!s->h_table[dst]) return fail(EINVAL, "destination slot not allocated");
	if ((r = check_packet_deep(packet, bytes, s->h_table, s->slot_bytes))) return r;
	if (set_device(s->dev)) return EIO;
	if ((r = ensure_dbk(s, n_mbs))) return r;
	E264Stream::Stage *st = &s->stage[s->stage_next];
	if (packet == st->h && bytes > st->cap) return fail(EINVAL, "packet larger than the buffer e264hip_packet_buffer returned");
	if (packet != st->h) { // caller did not use our pinned buffer: stage it
		void *h = e264hip_packet_buffer(s, bytes);
		if (!h) return ENOMEM;
		st = &s->stage[s->stage_next];
		memcpy(h, packet, bytes);
	}
	s->stage_next = (s->stage_next + 1) & 3;
	HIPCHK(hipMemcpyAsync(st->d, st->h, bytes, hipMemcpyHostToDevice, s->dev->q), EIO);
	// the job record rides at the tail of the pinned staging buffer's lifetime: tiny H2D on the same queue
	E264Job *job = (E264Job *)((uint8_t *)st->h + st->cap); // pinned, lives as long as the staging slot
	job->packet = st->d; job->dpb = s->d_table; job->dbk = s->d_dbk;
	HIPCHK(hipMemcpyAsync(st->d_job, job, sizeof(*job), hipMemcpyHostToDevice, s->dev->q), EIO);
	uint64_t serial = 0;
	r = launch(s->dev, st->d_job, 1, n_mbs, n_tiles, E264_RUN_ALL, &serial);
	if (r) return r;
	s->slot_serial[dst] = serial;
	hipEventRecord(st->done, s->dev->q);
	st->busy = true;
	return 0;
}

API int e264hip_frame_wait(E264Stream *s, int slot)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS) return fail(EINVAL, "frame_wait");
	if (!s->h_table[slot]) return fail(EINVAL, "frame_wait slot");
	if (set_device(s->dev)) return EIO;
	return wait_slot(s, slot);
}

API int e264hip_frame_download(E264Stream *s, int slot, void *dst, size_t bytes)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS || !s->h_table[slot]) return fail(EINVAL, "frame_download slot");
	if (set_device(s->dev)) return EIO;
	if (!dst) dst = s->mirror[slot];
	if (!dst) return fail(EINVAL, "no destination");
	if (bytes == 0 || bytes > s->slot_bytes[slot]) bytes = s->slot_bytes[slot];
	int r = wait_slot(s, slot); // the frame is final on the device; the copy runs beside whatever other decoders have queued since
	if (r) return r;
	hipStream_t qc = s->dev->qc ? s->dev->qc : s->dev->q;
	if (!s->dl_done && hipEventCreateWithFlags(&s->dl_done, E264_WAIT_EVENT) != hipSuccess) { s->dl_done = nullptr; return fail(EIO, "hipEventCreate"); }
	HIPCHK(hipMemcpyAsync(dst, s->h_table[slot], bytes, hipMemcpyDeviceToHost, qc), EIO);
	HIPCHK(hipEventRecord(s->dl_done, qc), EIO);
	HIPCHK(hipEventSynchronize(s->dl_done), EIO);
	return 0;
}

API int e264hip_packet_upload(E264Device *dev, const void *packet, size_t bytes, E264Packet **out)
{
	if (!dev || !out) return fail(EINVAL, "null argument");
	int dst, n_mbs, n_tiles, r = check_packet(packet, bytes, &dst, &n_mbs, &n_tiles);
	if (r) return r;
	uint32_t ref_mask = 0;
	if ((r = check_packet_deep(packet, bytes, nullptr, nullptr, &ref_mask))) return r;
	if (set_device(dev)) return EIO;
	E264Packet *p = new (std::nothrow) E264Packet();
	if (!p) return fail(ENOMEM, "packet object");
	p->dev = dev; p->bytes = bytes; p->dst_slot = dst; p->n_mbs = n_mbs; p->n_tiles = n_tiles; p->ref_mask = ref_mask;
	p->frame_bytes = (uint64_t)((const E264FrameHdr *)packet)->plane_size_Y + ((const E264FrameHdr *)packet)->plane_size_C;
	if (hipMalloc((void **)&p->d_bytes, bytes) != hipSuccess) { delete p; return fail(ENOMEM, "hipMalloc packet"); }
	hipError_t e = hipMemcpy(p->d_bytes, packet, bytes, hipMemcpyHostToDevice);
	if (e != hipSuccess) { hipFree(p->d_bytes); delete p; return fail(EIO, "hipMemcpy packet", e); }
	*out = p;
	return 0;
}

API void e264hip_packet_free(E264Packet *p)
{
	if (!p) return;
	hipSetDevice(p->dev->ordinal);
	hipStreamSynchronize(p->dev->q);
	hipFree(p->d_bytes);
	delete p;
}

struct E264Batch {
	E264Device *dev;
	E264Job *d_jobs;
	int n, max_mbs, max_tiles;
	std::vector<std::pair<E264Stream *, int>> writes; // (stream, destination slot) of every job
};

API int e264hip_batch_create(E264Device *dev, E264Stream *const *streams, E264Packet *const *packets, int n, E264Batch **out)
{
	if (!dev || !streams || !packets || !out || n <= 0) return fail(EINVAL, "batch_create arguments");
	if (set_device(dev)) return EIO;
	std::vector<E264Job> jobs((size_t)n);
	int max_mbs = 0, max_tiles = 0;
	for (int i = 0; i < n; i++) {
		if (!streams[i] || !packets[i] || streams[i]->dev != dev || packets[i]->dev != dev) return fail(EINVAL, "batch entry");
		if (!streams[i]->h_table[packets[i]->dst_slot]) return fail(EINVAL, "destination slot not allocated");
		// the packet was vetted without a stream at upload time: its slots against THIS stream's allocations
		if (packets[i]->frame_bytes > streams[i]->slot_bytes[packets[i]->dst_slot]) return fail(EINVAL, "picture larger than the destination slot");
		for (int sl = 0; sl < E264_MAX_SLOTS; sl++)
			if (packets[i]->ref_mask >> sl & 1) {
				if (!streams[i]->h_table[sl]) return fail(EINVAL, "reference slot not allocated");
				if (packets[i]->frame_bytes > streams[i]->slot_bytes[sl]) return fail(EINVAL, "picture larger than a reference slot");
			}
		for (int j = 0; j < i; j++) // two jobs of one stream would share its DPB and parameter buffer inside one launch
			if (streams[j] == streams[i]) return fail(EINVAL, "a stream may contribute one frame per batch");
		int r = ensure_dbk(streams[i], packets[i]->n_mbs);
		if (r) return r;
		jobs[i].packet = packets[i]->d_bytes;
		jobs[i].dpb = streams[i]->d_table;
		jobs[i].dbk = streams[i]->d_dbk;
		if (packets[i]->n_mbs > max_mbs) max_mbs = packets[i]->n_mbs;
		if (packets[i]->n_tiles > max_tiles) max_tiles = packets[i]->n_tiles;
	}
	E264Batch *b = new (std::nothrow) E264Batch();
	if (!b) return fail(ENOMEM, "batch object");
	b->dev = dev; b->n = n; b->max_mbs = max_mbs; b->max_tiles = max_tiles;
	for (int i = 0; i < n; i++) b->writes.emplace_back(streams[i], packets[i]->dst_slot);
	if (hipMalloc((void **)&b->d_jobs, sizeof(E264Job) * n) != hipSuccess) { delete b; return fail(ENOMEM, "hipMalloc jobs"); }
	hipError_t e = hipMemcpy(b->d_jobs, jobs.data(), sizeof(E264Job) * n, hipMemcpyHostToDevice);
	if (e != hipSuccess) { hipFree(b->d_jobs); delete b; return fail(EIO, "hipMemcpy jobs", e); }
	*out = b;
	return 0;
}

API int e264hip_batch_submit(E264Batch *b, int mode)
{
	if (!b) return fail(EINVAL, "null batch");
	if (set_device(b->dev)) return EIO;
	uint64_t serial = 0;
	int r = launch(b->dev, b->d_jobs, b->n, b->max_mbs, b->max_tiles, mode, &serial);
	if (!r) for (auto &w : b->writes) w.first->slot_serial[w.second] = serial;
	return r;
}

API void e264hip_batch_free(E264Batch *b)
{
	if (!b) return;
	hipSetDevice(b->dev->ordinal);
	hipStreamSynchronize(b->dev->q);
	hipFree(b->d_jobs);
	delete b;
}

API int e264hip_submit_batch(E264Device *dev, E264Stream *const *streams, E264Packet *const *packets, int n, int mode)
{
	E264Batch *b = nullptr;
	int r = e264hip_batch_create(dev, streams, packets, n, &b);
	if (r) return r;
	r = e264hip_batch_submit(b, mode);
	e264hip_batch_free(b); // synchronises the queue: convenience path for tests
	return r;
}

// Host packets of MANY streams, one submission, nothing synchronous: every packet is staged through its
// stream's pinned ring and copied on the device queue, the job table through a device-level ring, then the four
// kernels are launched.  The caller may reuse / free the host packets on return.
static int submit_host_impl(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode, int flags);
API int e264hip_submit_batch_host(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode)
{
	return submit_host_impl(dev, streams, packets, bytes, n, mode, 0);
}
// The same for packets that already sit in page-locked memory (e264hip_host_alloc; what a front end does when it assembles
// the packet in place): no staging copy, the H2D reads the caller's buffer, which must stay untouched until the submission
// has retired (e264hip_device_sync / a later e264hip_frame_wait).  E264_SUBMIT_TRUSTED: the caller has run
// e264hip_packet_check on exactly these bytes (a packet produced by its own emitter, a validated capture): the
// per-macroblock walk is not repeated on the submitting thread.
API int e264hip_submit_batch_pinned(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode, int flags)
{
	return submit_host_impl(dev, streams, packets, bytes, n, mode, 1 | (flags & E264_SUBMIT_TRUSTED ? 2 : 0));
}
API void *e264hip_host_alloc(E264Device *dev, size_t bytes)
{
	void *p = nullptr;
	if (!dev || set_device(dev) || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { fail(ENOMEM, "hipHostMalloc"); return nullptr; }
	return p;
}
API void e264hip_host_free(E264Device *dev, void *p)
{
	if (dev && p && !set_device(dev)) hipHostFree(p);
}
static int submit_host_impl(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode, int flags)
{
	const bool pinned = flags & 1, trusted = flags & 2;
	if (!dev || !streams || !packets || !bytes || n <= 0) return fail(EINVAL, "submit_batch_host arguments");
	if (set_device(dev)) return EIO;
	std::vector<int> mbs_of((size_t)n), tiles_of((size_t)n), rc((size_t)n, 0), dst_of((size_t)n);
	std::vector<std::string> why((size_t)n);
	for (int i = 0; i < n; i++) {
		if (!streams[i] || streams[i]->dev != dev) return fail(EINVAL, "batch entry");
		for (int j = 0; j < i; j++)
			if (streams[j] == streams[i]) return fail(EINVAL, "a stream may contribute one frame per batch");
	}
	{ // validate the whole batch before the first side effect: every packet on its own, in parallel
		std::lock_guard<std::mutex> pg(g_pool_user);
		g_pool.parallel_for(n, [&](int i) {
			E264Stream *s = streams[i];
			int dst, r = check_packet(packets[i], bytes[i], &dst, &mbs_of[i], &tiles_of[i]);
			dst_of[i] = dst;
			if (!r && !s->h_table[dst]) r = fail(EINVAL, "destination slot not allocated");
			if (!r && !trusted) r = check_packet_deep(packets[i], bytes[i], s->h_table, s->slot_bytes);
			if (!r && trusted && (uint64_t)((const E264FrameHdr *)packets[i])->plane_size_Y + ((const E264FrameHdr *)packets[i])->plane_size_C > s->slot_bytes[dst])
				r = fail(EINVAL, "picture larger than the destination slot");
			if (r) { rc[i] = r; why[i] = g_err; } // the message lives in the worker's thread-local buffer
		});
	}
	for (int i = 0; i < n; i++)
		if (rc[i]) return fail(rc[i], why[i].c_str());
	std::lock_guard<std::mutex> bg(dev->batch_lock); // batches of one device are serialised (their streams are disjoint per batch anyway)
	E264Device::JobRing &jr = dev->jring[dev->jring_next];
	dev->jring_next = (dev->jring_next + 1) & 3;
	if (jr.busy) { hipEventSynchronize(jr.done); jr.busy = false; }
	if (jr.cap < n) {
		if (jr.h) hipHostFree(jr.h);
		if (jr.d) hipFree(jr.d);
		jr.h = nullptr; jr.d = nullptr; jr.cap = 0;
		int cap = (n + 63) & ~63;
		if (hipHostMalloc((void **)&jr.h, sizeof(E264Job) * cap, hipHostMallocDefault) != hipSuccess) return fail(ENOMEM, "pinned job table");
		if (hipMalloc((void **)&jr.d, sizeof(E264Job) * cap) != hipSuccess) { hipHostFree(jr.h); jr.h = nullptr; return fail(ENOMEM, "device job table"); }
		jr.cap = cap;
		if (!jr.done) hipEventCreateWithFlags(&jr.done, E264_WAIT_EVENT);
	}
	int max_mbs = 0, max_tiles = 0;
	std::vector<E264Stream::Stage *> stage_of((size_t)n);
	for (int i = 0; i < n; i++) { // staging slots (HIP calls: this thread only)
		E264Stream *s = streams[i];
		const int n_mbs = mbs_of[i];
		if (tiles_of[i] > max_tiles) max_tiles = tiles_of[i];
		int r = ensure_dbk(s, n_mbs);
		if (r) return r;
		if (!e264hip_packet_buffer(s, bytes[i])) return ENOMEM;
		stage_of[i] = &s->stage[s->stage_next];
		s->stage_next = (s->stage_next + 1) & 3;
		if (n_mbs > max_mbs) max_mbs = n_mbs;
	}
	if (!pinned) {
		std::lock_guard<std::mutex> pg(g_pool_user);
		g_pool.parallel_for(n, [&](int i) { memcpy(stage_of[i]->h, packets[i], bytes[i]); });
	}
	for (int i = 0; i < n; i++) {
		E264Stream::Stage *st = stage_of[i];
		HIPCHK(hipMemcpyAsync(st->d, pinned ? packets[i] : st->h, bytes[i], hipMemcpyHostToDevice, dev->q), EIO);
		jr.h[i].packet = st->d; jr.h[i].dpb = streams[i]->d_table; jr.h[i].dbk = streams[i]->d_dbk;
	}
	HIPCHK(hipMemcpyAsync(jr.d, jr.h, sizeof(E264Job) * n, hipMemcpyHostToDevice, dev->q), EIO);
	uint64_t serial = 0;
	int r = launch(dev, jr.d, n, max_mbs, max_tiles, mode, &serial);
	if (r) return r;
	for (int i = 0; i < n; i++) streams[i]->slot_serial[dst_of[i]] = serial;
	hipEventRecord(jr.done, dev->q);
	jr.busy = true;
	for (int i = 0; i < n; i++) { // the staging slots are free again when this submission has retired
		E264Stream::Stage &st = streams[i]->stage[(streams[i]->stage_next + 3) & 3];
		hipEventRecord(st.done, dev->q);
		st.busy = true;
	}
	return 0;
}

API int e264hip_event_record(E264Device *dev, int idx)
{
	if (!dev || idx < 0 || idx >= 16) return fail(EINVAL, "event index");
	if (set_device(dev)) return EIO;
	HIPCHK(hipEventRecord(dev->ev[idx], dev->q), EIO);
	return 0;
}

API int e264hip_event_elapsed_ms(E264Device *dev, int a, int b, float *ms)
{
	if (!dev || !ms || a < 0 || a >= 16 || b < 0 || b >= 16) return fail(EINVAL, "event index");
	if (set_device(dev)) return EIO;
	HIPCHK(hipEventSynchronize(dev->ev[b]), EIO);
	HIPCHK(hipEventElapsedTime(ms, dev->ev[a], dev->ev[b]), EIO);
	return 0;
}

API int e264hip_kernel_timing(E264Device *dev, int enable)
{
	if (!dev) return fail(EINVAL, "null device");
	e264hip_device_sync(dev);
	dev->ktiming = enable != 0;
	dev->kev_used = 0;
	return 0;
}

API int e264hip_kernel_time_ms(E264Device *dev, double *ms4, int *launches)
{
	if (!dev || !ms4) return fail(EINVAL, "null argument");
	int r = e264hip_device_sync(dev);
	if (r) return r;
	ms4[0] = ms4[1] = ms4[2] = ms4[3] = 0;
	for (size_t i = 0; i < dev->kev_used; i++)
		for (int k = 0; k < 4; k++) {
			float ms = 0;
			const E264Device::Marks &m = dev->kev[i];
			const bool side = k == 0 && m.side; // the parameter kernel ran on the second queue: its own pair of events
			if (hipEventElapsedTime(&ms, side ? m.a[0] : m.e[k], side ? m.a[1] : m.e[k + 1]) == hipSuccess) ms4[k] += ms;
		}
	if (launches) *launches = (int)dev->kev_used;
	return 0;
}
