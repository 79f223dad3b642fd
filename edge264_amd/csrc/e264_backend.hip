// e264_backend.hip -- C-ABI of the MI355X reconstruction back end (include/edge264_hip.h).
//
// Host side of the drop-in boundary: owns the device DPB of every decoder ("stream"), moves
// command packets to HBM with pinned async copies, launches the frame kernels on the compute
// lanes (HIP queues) of the device, and copies finished frames back on demand.  No torch types, no CPU
// fallback: if there is no gfx950 device every entry point fails with ENODEV.
//
// Queues of one device:
//   q[lane]  compute lanes.  A stream is bound to one lane (e264hip_stream_bind_lane, default 0); everything that touches
//            its DPB is ordered there.  Submissions of streams on different lanes run beside each other on the GPU.
//   qup      upload queue: the packet and job-table H2D copies of host batches.  A lane waits for ITS batch's copy event
//            only, so batch n+1 is in flight over PCIe while the kernels of batch n run (round 2: both shared one in-order
//            queue and a submission cost copy time + kernel time).
//   qc       download queue (edge264_get_frame's D2H beside the kernels).
// Nothing a single decoder does (frame_alloc / frame_free / flush / close / get_frame) waits for the device: each waits
// for that decoder's own last submission at most; memory it gives back is parked and recycled (hipFree would drain every
// queue of the device).
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
#include "../../include/edge264_compact.h"

#define API extern "C" __attribute__((visibility("default")))

static thread_local char g_err[256];
static int fail(int code, const char *what, hipError_t e = hipSuccess)
{
	snprintf(g_err, sizeof(g_err), "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
	return code;
}
#define HIPCHK(call, code) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(code, #call, e_); } while (0)

API const char *e264hip_last_error(void) { return g_err; }
// Build-time switches of this library, space separated; "" for the product build.  An E264_ABL_* / E264_PHASE_* entry means the
// kernels compute WRONG SAMPLES on purpose (timing ablations): such a library only opens a device with E264_ALLOW_ABLATION=1.
API const char *e264hip_build_flags(void) { return e264_kernel_build_flags(); }
static bool ablation_build_refused()
{
	const char *f = e264_kernel_build_flags();
	if (!strstr(f, "E264_ABL_") && !strstr(f, "E264_PHASE_")) return false;
	const char *a = getenv("E264_ALLOW_ABLATION");
	return !(a && a[0] == '1');
}
// events a host thread waits on: blocking, so that the waiter sleeps instead of spinning on a core the parser threads could use
#define E264_WAIT_EVENT (hipEventDisableTiming | hipEventBlockingSync)

// A few host threads for the per-packet work of a batch that arrives in ordinary host memory (validation of every
// macroblock record + the copy into page-locked staging memory: 0.18 ms per 1080p packet on one thread = 5 k frames/s,
// while PCIe carries 28 k).  No HIP call is ever made from these threads.  E264_HOST_THREADS overrides the count (0: none).
// One pool per device: the submitter threads of several GPUs (e264_multi --devices) do not queue behind each other.
namespace {
struct HostPool {
	std::vector<std::thread> th;
	std::mutex m;
	std::condition_variable cv, done_cv;
	const std::function<void(int)> *fn = nullptr;
	std::atomic<int> next{0};
	int n = 0, active = 0, limit = 0;
	uint64_t gen = 0;
	bool stop = false, started = false;
	void run() { for (int i; (i = next.fetch_add(1)) < n;) (*fn)(i); }
	void worker(int id)
	{
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lk(m);
		for (;;) {
			cv.wait(lk, [&] { return stop || gen != seen; });
			if (stop) return;
			seen = gen;
			const bool mine = id < limit; // (a job may ask for fewer workers than the pool has)
			lk.unlock();
			if (mine) run();
			lk.lock();
			if (--active == 0) done_cv.notify_one();
		}
	}
	// max_workers: pool threads that take part beside the caller (0: all).  Items that only copy (a trusted batch's gather into the staging buffer) are bound by
	// memory, not by cores: one buffer per stream (what a front end leaves) 106 / 109 / 108 k frames/s with 8 / 12 / 15 workers, four too few (66.7 k on one box);
	// tools/pin_probe.py, profiles/r06_ablations.txt item 17
	void parallel_for(int count, const std::function<void(int)> &f, int max_workers = 0)
	{
		std::unique_lock<std::mutex> lk(m);
		if (!started) {
			started = true;
			const char *e = getenv("E264_HOST_THREADS");
			int want = e ? atoi(e) : (int)std::min(15u, std::thread::hardware_concurrency() / 2);
			for (int i = 0; i < want; i++) th.emplace_back([this, i] { worker(i); });
		}
		if (th.empty() || count < 4) { lk.unlock(); for (int i = 0; i < count; i++) f(i); return; }
		fn = &f; n = count; next = 0; active = (int)th.size(); gen++;
		limit = max_workers > 0 ? max_workers : (int)th.size();
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
}

struct E264Packet {
	E264Device *dev;
	uint8_t *d_bytes;
	size_t bytes;
	int dst_slot;
	int n_mbs, n_tiles;
	uint64_t frame_bytes;  // plane_size_Y + plane_size_C the kernels will touch in every slot the packet names
	uint32_t ref_mask;     // DPB slots its motion refers to
	bool pred_work;        // it holds inter or PCM macroblocks (else e264_pred_kernel has nothing to do for it)
	bool has_l1;           // some macroblock predicts from list 1 (else the parameter kernel's small form will do)
};

#define E264_GATHER_WORKERS 12 // pool threads that gather a trusted batch into its staging buffer (copy only: see HostPool::parallel_for; E264_GATHER_THREADS overrides)
#define E264_JOB_RING 4 // batches in flight per device: one uploading, one in the kernels, one retiring
struct E264Device {
	int ordinal;
	enum { NQ = E264_MAX_LANES };
	hipStream_t q[NQ];         // compute lanes
	hipStream_t qup = nullptr; // upload queue
	hipStream_t qc = nullptr;  // download queue
	int waves;                 // waves per frame workgroup of the deblocking kernel (5 macroblock rows each): 2, 4, 7 or 8
	int intra_waves;           // waves per frame workgroup of the intra kernel (1 macroblock row each)
	std::atomic<int> max_lane{0}; // highest lane a stream was ever bound to
	int n_cus;                 // compute units of the device
	int split_planes;          // option "split_planes" (default 1): the split-off pictures' intra pass with luma and chroma on two workgroups (e264_intra_planes_kernel)
	int split_intra;           // option "split_intra" (default 1): in a submission that mixes I pictures with others, their intra pass runs on q2 from the start (E264Fork.n_nopred)
	int side_queue;            // option "side_queue": parameter kernel on a second queue beside the macroblock-parallel kernel
	int upload_queue;          // option "upload_queue" (default 1): the H2D copies of host batches go through qup
	hipStream_t q2[NQ];        // one second queue per compute lane (+ its fork / join events): lanes must not share one (their split submissions would queue behind each other)
	hipEvent_t forked[NQ], joined[NQ];
	hipEvent_t lane_ev[NQ];    // e264hip_event_record: joins the other lanes into lane 0
	std::mutex lock;           // kernel launches + their timing marks + the submission event ring
	std::mutex batch_lock;     // host batches: job ring, staging
	hipEvent_t ev[16];
	// per-launch kernel timing
	bool ktiming;
	struct Marks { hipEvent_t e[5], a[2]; bool side; };
	std::vector<Marks> kev;
	size_t kev_used;
	// job tables of host-packet batches: a ring of pinned + device buffers, each with the event of its upload and of its kernels
	struct JobRing {
		E264Job *h = nullptr, *d = nullptr; int cap = 0; hipEvent_t done = nullptr, up = nullptr; bool busy = false;
		// pageable batches: the packets of the WHOLE batch back to back in one page-locked buffer and one device buffer, so that a
		// batch crosses PCIe as ONE transfer (256 copies of 1 MB each reached 35 GB/s, and cost the submitting thread 257 driver calls)
		uint8_t *ph = nullptr, *pd = nullptr; size_t pcap = 0;
		// wire packets of the batch (include/edge264_compact.h) are unfolded HERE, by e264_expand_kernel on the upload queue right behind the batch's copy --
		// beside the kernels of the batch before, whose streams read THEIR ring slot's buffer: nothing to order but what the ring already orders
		uint8_t *xd = nullptr; size_t xcap = 0;
	} jring[E264_JOB_RING];
	int jring_next = 0;
	// Which submission wrote a slot last, and when it has retired: edge264_get_frame of ONE decoder must not wait for the
	// whole device (every later batch of every other decoder), only for the submission that produced its frame.
	uint64_t serial = 0;       // submissions so far (guarded by `lock`)
	enum { NEV = 1024 };       // (round 5: 64 wrapped at once when hundreds of decoders filled their slots -- every fill takes a serial -- and every wait fell back to the whole lane)
	hipEvent_t sub_ev[NEV] = {};
	uint64_t sub_serial[NEV] = {};
	int sub_lane[NEV] = {};
	std::atomic<uint64_t> lane_retired[NQ] = {}; // per lane: the highest serial anybody has SEEN retired (a lane runs in order: everything below has too)
	// Memory recycler.  hipFree / hipHostFree drain EVERY queue of the device, so what a decoder gives back (frame slots and
	// their host mirrors at an SPS change or edge264_free, parameter and staging buffers) is parked with the submission that
	// may still read it and handed to the next request of the same size once that submission has retired.  Parked memory is
	// released for real when the device closes (or when more than PARK_MAX blocks wait: the oldest retired ones go).
	struct Parked { void *p; size_t bytes; uint64_t serial; int lane; bool host; };
	enum { PARK_MAX = 4096 };
	std::vector<Parked> parked;
	std::mutex park_lock;
	HostPool pool;             // validation + staging copies of this device's host batches
	std::mutex pool_user;      // one batch at a time uses the pool
};

struct E264Stream {
	E264Device *dev;
	int lane;                             // compute lane of the device everything of this stream is ordered on
	uint8_t **d_table;                    // device array [E264_MAX_SLOTS] of slot pointers
	uint8_t *h_table[E264_MAX_SLOTS];     // same, host copy
	void *mirror[E264_MAX_SLOTS];         // pinned host mirrors
	size_t slot_bytes[E264_MAX_SLOTS];
	uint8_t *d_dbk;                       // per-stream scratch of the kernels: E264_SCRATCH_BYTES(dbk_mbs) (deblocking parameters + the intra bitmap)
	size_t dbk_mbs;
	uint8_t *d_expand;                    // where e264_expand_kernel unfolds the wire packets (include/edge264_compact.h) of e264hip_frame_submit, expand_cap bytes; NULL until the first one (batches: the ring slot's buffer)
	size_t expand_cap;
	// the slot table reaches the device from a small pinned ring (asynchronous: a pageable source would make hipMemcpyAsync
	// wait for the queue)
	enum { NTAB = 4 };
	uint8_t **tab_pin;                    // [NTAB][E264_MAX_SLOTS]
	hipEvent_t tab_ev[NTAB];
	bool tab_busy[NTAB];
	int tab_next;
	// packet staging ring (pinned host) + device copies
	struct Stage { void *h; uint8_t *d; size_t cap; hipEvent_t done; bool busy; E264Job *d_job; } stage[4];
	int stage_next;
	// Written by the decoder's own thread (fills, frees) AND by the thread that submits its batches (e264_multi: another one), hence
	// atomic and only ever raised: serials are handed out in queue order, so "the latest thing queued for this stream" is a maximum,
	// whichever thread's bookkeeping lands last.  (Round 3 kept a `loose` flag beside a plain serial: a submitter's late "no fill
	// pending" could erase a parser's "fill pending" and a freed slot went to another lane with the fill still queued.)
	std::atomic<uint64_t> slot_serial[E264_MAX_SLOTS]; // what wrote the slot last: a submission or a fill (0: nothing since it was allocated)
	std::atomic<uint64_t> last_serial;                 // the latest submission or fill of the stream
	hipEvent_t dl_done;                   // the stream's last download
};

static uint64_t mark_lane(E264Device *dev, int lane);
static void raise_serial(std::atomic<uint64_t> &a, uint64_t v)
{
	uint64_t cur = a.load(std::memory_order_relaxed);
	while (cur < v && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
}
static int set_device(E264Device *dev)
{
	HIPCHK(hipSetDevice(dev->ordinal), EIO);
	return 0;
}
static hipStream_t lane_of(const E264Stream *s) { return s->dev->q[s->lane]; }

// ---------------------------------------------------------------------------------------------------------------------
// submissions: retired? / wait
// ---------------------------------------------------------------------------------------------------------------------
static hipEvent_t serial_event(E264Device *dev, uint64_t serial)
{
	if (!serial) return nullptr;
	std::lock_guard<std::mutex> g(dev->lock);
	const int idx = (int)(serial % E264Device::NEV);
	return dev->sub_serial[idx] == serial ? dev->sub_ev[idx] : nullptr;
}
// Has submission `serial` of lane `lane` left the GPU?  (serial 0: nothing was ever submitted)
static bool serial_retired(E264Device *dev, uint64_t serial, int lane)
{
	if (!serial) return true;
	std::atomic<uint64_t> &seen = dev->lane_retired[lane];
	if (serial <= seen.load(std::memory_order_relaxed)) return true; // no lock, no driver call: somebody saw this lane get past it
	// An event handle read under dev->lock is queried OUTSIDE it, and ring entry serial % NEV is the next one launch() / mark_lane() record
	// again -- possibly for ANOTHER lane: a "done" answer only counts if the entry still belongs to the serial it was read for once the query
	// has returned (checked under the lock again); otherwise the answer is about somebody else's marker and nothing may be cached from it.
	auto still = [&](uint64_t want) {
		std::lock_guard<std::mutex> g(dev->lock);
		const int idx = (int)(want % E264Device::NEV);
		return dev->sub_serial[idx] == want && dev->sub_lane[idx] == lane;
	};
	hipEvent_t ev = serial_event(dev, serial);
	if (ev) {
		if (hipEventQuery(ev) != hipSuccess) { (void)hipGetLastError(); return false; }
		if (still(serial)) { raise_serial(seen, serial); return true; }
		// re-recorded while we asked: the ring has wrapped past this serial, fall through to the wrapped path
	}
	// the event ring has wrapped past this serial.  A lane runs its work in order: a NEWER entry of the same lane that has retired proves
	// this one has (a continuously busy lane is never idle, hipStreamQuery alone would keep old blocks parked for good).  ONE query: the
	// oldest such entry still in the ring, picked under the lock, asked outside it (round 4 asked up to 64 under the lock, per parked block).
	hipEvent_t probe = nullptr;
	uint64_t probe_serial = 0;
	{
		std::lock_guard<std::mutex> g(dev->lock);
		for (int i = 0; i < E264Device::NEV; i++)
			if (dev->sub_serial[i] > serial && dev->sub_lane[i] == lane && dev->sub_ev[i] && (!probe || dev->sub_serial[i] < probe_serial)) { probe = dev->sub_ev[i]; probe_serial = dev->sub_serial[i]; }
	}
	if (probe && hipEventQuery(probe) == hipSuccess && still(probe_serial)) { raise_serial(seen, probe_serial); return true; }
	(void)hipGetLastError(); // (hipErrorNotReady is not an error to keep)
	return hipStreamQuery(dev->q[lane]) == hipSuccess;
}
// Blocks until it has: the submission's own event, else (ring wrapped) the lane -- never the device.
static int serial_wait(E264Device *dev, uint64_t serial, int lane)
{
	if (!serial) return 0;
	hipEvent_t ev = serial_event(dev, serial);
	if (ev) {
		HIPCHK(hipEventSynchronize(ev), EIO);
		std::lock_guard<std::mutex> g(dev->lock); // (as in serial_retired: the entry may have been recorded again, for another lane, while we waited)
		const int idx = (int)(serial % E264Device::NEV);
		if (dev->sub_serial[idx] == serial && dev->sub_lane[idx] == lane) return 0;
	}
	HIPCHK(hipStreamSynchronize(dev->q[lane]), EIO);
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// memory recycler
// ---------------------------------------------------------------------------------------------------------------------
static void *mem_acquire(E264Device *dev, size_t bytes, bool host)
{
	{
		std::lock_guard<std::mutex> g(dev->park_lock);
		for (size_t i = 0; i < dev->parked.size(); i++) {
			E264Device::Parked &k = dev->parked[i];
			if (k.host == host && k.bytes == bytes && serial_retired(dev, k.serial, k.lane)) {
				void *p = k.p;
				dev->parked.erase(dev->parked.begin() + (ptrdiff_t)i);
				return p;
			}
		}
	}
	void *p = nullptr;
	hipError_t e = host ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
	if (e == hipSuccess) return p;
	// out of memory with blocks of other sizes parked (SPS size changes, many open / close cycles): the retired ones go back to the
	// driver (this drains the device's queues: the price of running out), then once more
	(void)hipGetLastError();
	std::vector<E264Device::Parked> drop;
	{
		std::lock_guard<std::mutex> g(dev->park_lock);
		for (size_t i = 0; i < dev->parked.size();) {
			if (serial_retired(dev, dev->parked[i].serial, dev->parked[i].lane)) { drop.push_back(dev->parked[i]); dev->parked.erase(dev->parked.begin() + (ptrdiff_t)i); }
			else i++;
		}
	}
	if (drop.empty()) return nullptr;
	for (auto &k : drop) { if (k.host) hipHostFree(k.p); else hipFree(k.p); }
	p = nullptr;
	e = host ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
	if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return p;
}
// `serial` / `lane`: the last submission that may touch the block (0: none)
static void mem_release(E264Device *dev, void *p, size_t bytes, bool host, uint64_t serial, int lane)
{
	if (!p) return;
	std::vector<E264Device::Parked> drop;
	{
		std::lock_guard<std::mutex> g(dev->park_lock);
		dev->parked.push_back({p, bytes, serial, lane, host});
		if (dev->parked.size() > E264Device::PARK_MAX) { // the oldest retired blocks really go (this drains the device: rare)
			for (size_t i = 0; i < dev->parked.size() && drop.size() < E264Device::PARK_MAX / 4;) {
				if (serial_retired(dev, dev->parked[i].serial, dev->parked[i].lane)) { drop.push_back(dev->parked[i]); dev->parked.erase(dev->parked.begin() + (ptrdiff_t)i); }
				else i++;
			}
		}
	}
	for (auto &k : drop) { if (k.host) hipHostFree(k.p); else hipFree(k.p); }
}

API int e264hip_device_open(int ordinal, E264Device **out)
{
	if (!out) return fail(EINVAL, "null out");
	if (ablation_build_refused()) {
		snprintf(g_err, sizeof(g_err), "this library is a timing-ablation build (%s): its samples are wrong by design; set E264_ALLOW_ABLATION=1 to use it", e264_kernel_build_flags());
		return ENOTSUP;
	}
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
	d->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	d->waves = 108; // round 4: e264_deblock2_kernel, 8 waves that take luma groups (8 rows) and chroma groups (15 rows) from one list: 0.99 - 1.02 ms against
	// 1.02 - 1.03 for the mixed waves below (profiles/r04_ablations.txt item 7).  Before (waves = 8, e264_deblock_kernel): 40 macroblock rows in flight (all the LDS takes: 154 KB); since the samples are fetched four macroblocks at a time 8 waves beat
	              // 7 (1.081 -> 1.051 ms per 256 x 1080p; round 2, one macroblock per fetch: 7 was the optimum)
	d->intra_waves = 16; // 16 rows in flight: 1.6 -> 1.1 ms per 256-frame launch (the intra kernel fits 128 VGPRs)
	d->ktiming = false; d->kev_used = 0;
	d->upload_queue = 1;
	for (int i = 0; i < E264Device::NQ; i++) { d->q[i] = nullptr; d->lane_ev[i] = nullptr; }
	if (hipSetDevice(ordinal) != hipSuccess) { delete d; return fail(EIO, "hipSetDevice"); }
	for (int i = 0; i < E264Device::NQ; i++)
		if (hipStreamCreateWithFlags(&d->q[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d->lane_ev[i], hipEventDisableTiming) != hipSuccess) {
			for (int k = 0; k <= i; k++) { if (d->q[k]) hipStreamDestroy(d->q[k]); if (d->lane_ev[k]) hipEventDestroy(d->lane_ev[k]); }
			delete d;
			return fail(EIO, "hipStreamCreate");
		}
	for (int i = 0; i < 16; i++)
		hipEventCreate(&d->ev[i]);
	d->split_intra = 1; d->split_planes = 1;
	d->side_queue = 0; // measured: no gain (profiles/r01k_ablation_breakdown.txt), off by default
	for (int i = 0; i < E264Device::NQ; i++) { d->q2[i] = nullptr; d->forked[i] = d->joined[i] = nullptr; }
	// The runtime deals its hardware queues (four by default, GPU_MAX_HW_QUEUES) to HIP streams in the order they are made: lane 0's second queue, the download
	// queue and the upload queue come right after the lanes, as in every round so far; the other lanes' second queues LAST (made in front of qc / qup they pushed the
	// download queue onto lane 0's hardware queue: one stream through edge264.h 1.03 -> 1.62 ms per picture, gpurun_out/r06f)
	auto make_q2 = [&](int i) {
		return hipStreamCreateWithFlags(&d->q2[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&d->forked[i], hipEventDisableTiming) == hipSuccess &&
			hipEventCreateWithFlags(&d->joined[i], hipEventDisableTiming) == hipSuccess;
	};
	bool q2_ok = make_q2(0);
	if (hipStreamCreateWithFlags(&d->qc, hipStreamNonBlocking) != hipSuccess) d->qc = nullptr; // downloads then share the lane
	if (hipStreamCreateWithFlags(&d->qup, hipStreamNonBlocking) != hipSuccess) d->qup = nullptr; // uploads then share the lane
	for (int i = 1; i < E264Device::NQ && q2_ok; i++) q2_ok = make_q2(i);
	if (!q2_ok) // not fatal: the options that need a second queue then stay off
		for (int i = 0; i < E264Device::NQ; i++) {
			if (d->q2[i]) hipStreamDestroy(d->q2[i]);
			if (d->forked[i]) hipEventDestroy(d->forked[i]);
			if (d->joined[i]) hipEventDestroy(d->joined[i]);
			d->q2[i] = nullptr; d->forked[i] = d->joined[i] = nullptr;
		}
	for (int i = 0; i < E264Device::NEV; i++)
		if (hipEventCreateWithFlags(&d->sub_ev[i], E264_WAIT_EVENT) != hipSuccess) d->sub_ev[i] = nullptr;
	*out = d;
	return 0;
}

API int e264hip_device_sync(E264Device *dev)
{
	if (!dev) return fail(EINVAL, "null device");
	if (set_device(dev)) return EIO;
	if (dev->qup) HIPCHK(hipStreamSynchronize(dev->qup), EIO);
	for (int i = 0; i < E264Device::NQ; i++)
		HIPCHK(hipStreamSynchronize(dev->q[i]), EIO);
	for (int i = 0; i < E264Device::NQ; i++)
		if (dev->q2[i]) HIPCHK(hipStreamSynchronize(dev->q2[i]), EIO);
	return 0;
}

API void e264hip_device_close(E264Device *dev)
{
	if (!dev) return;
	hipSetDevice(dev->ordinal);
	e264hip_device_sync(dev);
	for (int i = 0; i < 16; i++) hipEventDestroy(dev->ev[i]);
	for (auto &p : dev->kev) { for (int i = 0; i < 5; i++) hipEventDestroy(p.e[i]); for (int i = 0; i < 2; i++) hipEventDestroy(p.a[i]); }
	for (int i = 0; i < E264Device::NQ; i++) {
		if (dev->q2[i]) hipStreamDestroy(dev->q2[i]);
		if (dev->forked[i]) hipEventDestroy(dev->forked[i]);
		if (dev->joined[i]) hipEventDestroy(dev->joined[i]);
	}
	for (auto &jr : dev->jring) {
		if (jr.h) hipHostFree(jr.h);
		if (jr.d) hipFree(jr.d);
		if (jr.ph) hipHostFree(jr.ph);
		if (jr.pd) hipFree(jr.pd);
		if (jr.xd) hipFree(jr.xd);
		if (jr.done) hipEventDestroy(jr.done);
		if (jr.up) hipEventDestroy(jr.up);
	}
	for (auto &k : dev->parked) { if (k.host) hipHostFree(k.p); else hipFree(k.p); }
	dev->parked.clear();
	if (dev->qc) { hipStreamSynchronize(dev->qc); hipStreamDestroy(dev->qc); }
	if (dev->qup) hipStreamDestroy(dev->qup);
	for (int i = 0; i < E264Device::NEV; i++) if (dev->sub_ev[i]) hipEventDestroy(dev->sub_ev[i]);
	for (int i = 0; i < E264Device::NQ; i++) { hipStreamDestroy(dev->q[i]); hipEventDestroy(dev->lane_ev[i]); }
	delete dev;
}

API int e264hip_set_option(E264Device *dev, const char *name, int value)
{
	if (!dev || !name) return -1;
	if (!strcmp(name, "split_planes")) {
		int prev = dev->split_planes;
		dev->split_planes = value != 0;
		return prev;
	}
	if (!strcmp(name, "split_intra")) {
		int prev = dev->split_intra;
		dev->split_intra = value == 2 ? 2 : value != 0; // 2: also with more than two lanes in use (for runs with GPU_MAX_HW_QUEUES raised)
		return prev;
	}
	if (!strcmp(name, "side_queue")) {
		int prev = dev->side_queue;
		dev->side_queue = dev->q2[0] ? (value == 1 || value == 2 ? value : 0) : 0; // 1: beside the prediction kernel, 2: beside the intra kernel
		return prev;
	}
	if (!strcmp(name, "upload_queue")) {
		int prev = dev->upload_queue;
		dev->upload_queue = value != 0;
		return prev;
	}
	if (!strcmp(name, "intra_waves")) {
		int prev = dev->intra_waves;
		if (value == 4 || value == 8 || value == 16) dev->intra_waves = value;
		return prev;
	}
	if (!strcmp(name, "waves")) {
		int prev = dev->waves;
		// 100 + n: n luma / chroma waves (e264_deblock2_kernel); 110, 112 exist only in builds with strips of four macroblocks (E264_DBK_GS = 2):
		// elsewhere they are refused (-1, setting kept) instead of silently running as 108
		const bool gs2 = strstr(e264_kernel_build_flags(), "E264_DBK_GS=2") != nullptr;
		if ((value == 110 || value == 112) && !gs2) return -1;
		if (value == 2 || value == 4 || value == 7 || value == 8 || value == 106 || value == 107 || value == 108 || value == 110 || value == 112) dev->waves = value; // anything else keeps the setting
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
	s->d_table = (uint8_t **)mem_acquire(dev, sizeof(uint8_t *) * E264_MAX_SLOTS, false);
	s->tab_pin = (uint8_t **)mem_acquire(dev, sizeof(uint8_t *) * E264_MAX_SLOTS * E264Stream::NTAB, true);
	if (!s->d_table || !s->tab_pin) {
		mem_release(dev, s->d_table, sizeof(uint8_t *) * E264_MAX_SLOTS, false, 0, 0);
		mem_release(dev, s->tab_pin, sizeof(uint8_t *) * E264_MAX_SLOTS * E264Stream::NTAB, true, 0, 0);
		delete s;
		return fail(ENOMEM, "slot table");
	}
	hipMemsetAsync(s->d_table, 0, sizeof(uint8_t *) * E264_MAX_SLOTS, lane_of(s));
	*out = s;
	return 0;
}

API int e264hip_stream_bind_lane(E264Stream *s, int lane)
{
	if (!s || lane < 0 || lane >= E264Device::NQ) return fail(EINVAL, "stream_bind_lane");
	if (lane == s->lane) return 0;
	if (set_device(s->dev)) return EIO;
	// what the old lane still has queued for this stream (table updates, fills, submissions) must be over before the new one starts
	HIPCHK(hipStreamSynchronize(lane_of(s)), EIO);
	s->lane = lane;
	int seen = s->dev->max_lane.load(std::memory_order_relaxed);
	while (seen < lane && !s->dev->max_lane.compare_exchange_weak(seen, lane, std::memory_order_relaxed)) {}
	return 0;
}

// Waits for what THIS stream has submitted (edge264_flush, src/edge264.c:261-270), not for the device.
API int e264hip_stream_flush(E264Stream *s)
{
	if (!s) return fail(EINVAL, "null stream");
	if (set_device(s->dev)) return EIO;
	if (!s->last_serial) { HIPCHK(hipStreamSynchronize(lane_of(s)), EIO); return 0; } // fills / uploads only
	return serial_wait(s->dev, s->last_serial, s->lane);
}

API void e264hip_stream_close(E264Stream *s)
{
	if (!s) return;
	E264Device *dev = s->dev;
	set_device(dev);
	// everything THIS stream has queued on its lane (submissions, fills, table updates) -- not the device, not the other lanes
	hipEvent_t end = nullptr;
	if (hipEventCreateWithFlags(&end, E264_WAIT_EVENT) == hipSuccess && hipEventRecord(end, lane_of(s)) == hipSuccess) hipEventSynchronize(end);
	else hipStreamSynchronize(lane_of(s));
	if (end) hipEventDestroy(end);
	for (int i = 0; i < E264_MAX_SLOTS; i++) {
		mem_release(dev, s->h_table[i], s->slot_bytes[i] + 64, false, 0, 0);
		mem_release(dev, s->mirror[i], s->slot_bytes[i], true, 0, 0);
	}
	for (auto &st : s->stage) {
		mem_release(dev, st.h, st.cap + 64, true, 0, 0);
		mem_release(dev, st.d, st.cap, false, 0, 0);
		mem_release(dev, st.d_job, sizeof(E264Job), false, 0, 0);
		if (st.done) hipEventDestroy(st.done);
	}
	mem_release(dev, s->d_dbk, E264_SCRATCH_BYTES(s->dbk_mbs), false, 0, 0);
	mem_release(dev, s->d_expand, s->expand_cap, false, 0, 0);
	if (s->dl_done) hipEventDestroy(s->dl_done);
	for (int i = 0; i < E264Stream::NTAB; i++)
		if (s->tab_ev[i]) hipEventDestroy(s->tab_ev[i]);
	mem_release(dev, s->tab_pin, sizeof(uint8_t *) * E264_MAX_SLOTS * E264Stream::NTAB, true, 0, 0);
	mem_release(dev, s->d_table, sizeof(uint8_t *) * E264_MAX_SLOTS, false, 0, 0);
	delete s;
}

// The slot table follows the stream's lane: enqueued after every kernel that still reads the old one, before any that
// needs the new one.  Asynchronous (pinned ring); the host waits only if it laps its own ring.
static int push_table(E264Stream *s)
{
	const int k = s->tab_next;
	s->tab_next = (k + 1) % E264Stream::NTAB;
	if (!s->tab_ev[k] && hipEventCreateWithFlags(&s->tab_ev[k], E264_WAIT_EVENT) != hipSuccess) { s->tab_ev[k] = nullptr; return fail(EIO, "hipEventCreate"); }
	if (s->tab_busy[k]) { HIPCHK(hipEventSynchronize(s->tab_ev[k]), EIO); s->tab_busy[k] = false; }
	uint8_t **pin = s->tab_pin + (size_t)k * E264_MAX_SLOTS;
	memcpy(pin, s->h_table, sizeof(uint8_t *) * E264_MAX_SLOTS);
	HIPCHK(hipMemcpyAsync(s->d_table, pin, sizeof(uint8_t *) * E264_MAX_SLOTS, hipMemcpyHostToDevice, lane_of(s)), EIO);
	HIPCHK(hipEventRecord(s->tab_ev[k], lane_of(s)), EIO);
	s->tab_busy[k] = true;
	return 0;
}

API int e264hip_frame_alloc(E264Stream *s, int slot, size_t samples_bytes, void **host_mirror)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS || samples_bytes == 0) return fail(EINVAL, "frame_alloc arguments");
	if (set_device(s->dev)) return EIO;
	if (s->h_table[slot]) e264hip_frame_free(s, slot);
	// +64: the MC fast path reads whole dwords up to 3 bytes past a 9-sample span (like the
	// reference's +16 over-read margin, src/edge264_headers.c:115)
	s->h_table[slot] = (uint8_t *)mem_acquire(s->dev, samples_bytes + 64, false);
	if (!s->h_table[slot]) return fail(ENOMEM, "hipMalloc frame");
	s->slot_bytes[slot] = samples_bytes;
	s->slot_serial[slot] = 0;
	if (host_mirror) {
		s->mirror[slot] = mem_acquire(s->dev, samples_bytes, true);
		if (!s->mirror[slot]) {
			mem_release(s->dev, s->h_table[slot], samples_bytes + 64, false, 0, 0); s->h_table[slot] = nullptr;
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
	// A host batch naming this stream may be under assembly on another thread (e264_multi's submitter: it has read the stream's slot pointers for the
	// validation and not launched yet): the free waits for that batch to be launched or refused ...
	std::lock_guard<std::mutex> bg(s->dev->batch_lock);
	// ... and kernels and fills already queued may still read or write the slot: it is parked until the stream's latest work has retired
	// behind the lane's CURRENT tail (a marker of its own), not s->last_serial: a batch the submitter thread has just launched may not have
	// published its serial to the stream yet (the block could be recycled while that batch still used it)
	mem_release(s->dev, s->h_table[slot], s->slot_bytes[slot] + 64, false, mark_lane(s->dev, s->lane), s->lane);
	mem_release(s->dev, s->mirror[slot], s->slot_bytes[slot], true, 0, 0); // downloads are synchronous: nothing in flight
	s->h_table[slot] = nullptr; s->mirror[slot] = nullptr; s->slot_bytes[slot] = 0; s->slot_serial[slot] = 0;
	push_table(s);
}

API void *e264hip_frame_device_ptr(E264Stream *s, int slot)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS) return nullptr;
	return s->h_table[slot];
}

API int e264hip_frame_fill(E264Stream *s, int slot, int value)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS || !s->h_table[slot]) return fail(EINVAL, "frame_fill slot");
	if (set_device(s->dev)) return EIO;
	HIPCHK(hipMemsetAsync(s->h_table[slot], value, s->slot_bytes[slot], lane_of(s)), EIO);
	const uint64_t serial = mark_lane(s->dev, s->lane); // the fill has a serial and an event of its own
	raise_serial(s->slot_serial[slot], serial);
	raise_serial(s->last_serial, serial);
	return 0;
}

API int e264hip_frame_upload(E264Stream *s, int slot, const void *src, size_t bytes)
{
	if (!s || slot < 0 || slot >= E264_MAX_SLOTS || !s->h_table[slot] || bytes > s->slot_bytes[slot]) return fail(EINVAL, "frame_upload");
	if (set_device(s->dev)) return EIO;
	HIPCHK(hipMemcpyAsync(s->h_table[slot], src, bytes, hipMemcpyHostToDevice, lane_of(s)), EIO);
	HIPCHK(hipStreamSynchronize(lane_of(s)), EIO);
	s->slot_serial[slot] = 0;
	return 0;
}

// tiles: workgroups e264_pred_kernel needs for this frame
// area (may be null): 0 for a version-4 packet; for a WIRE packet (version 5, include/edge264_compact.h), whose structure is checked here,
// the bytes its expansion on the device needs
static int check_packet(const void *packet, size_t bytes, int *dst, int *n_mbs, int *tiles = nullptr, size_t *area = nullptr)
{
	const E264FrameHdr *h = (const E264FrameHdr *)packet;
	if (!packet || bytes < sizeof(*h) || h->magic != E264_MAGIC || (h->version != E264_VERSION && h->version != E264_VERSION_COMPACT) || h->total_bytes > bytes)
		return fail(EINVAL, "not a command packet");
	if (h->dst_slot < 0 || h->dst_slot >= E264_MAX_SLOTS) return fail(EINVAL, "dst_slot");
	if (area) *area = 0;
	if (h->version == E264_VERSION_COMPACT) {
		if (e264_check_compact(packet, h->total_bytes)) return fail(EINVAL, "wire packet structure");
		if (area) *area = e264_expand_area_bytes(packet);
		*dst = h->dst_slot;
		*n_mbs = (int)h->width_mbs * h->height_mbs;
		if (tiles) *tiles = e264_pred_tiles(h->width_mbs, h->height_mbs);
		return 0;
	}
	size_t n_mb = (size_t)h->width_mbs * h->height_mbs;
	size_t need = (size_t)h->mbs_off + n_mb * sizeof(E264Mb);
	if (h->motion_off && (h->motion_off < need || (need = (size_t)h->motion_off) > h->total_bytes)) return fail(EINVAL, "motion section");
	if (need > h->payload_off || (size_t)h->payload_off + h->payload_bytes > h->total_bytes) return fail(EINVAL, "packet layout");
	*dst = h->dst_slot;
	*n_mbs = (int)h->width_mbs * h->height_mbs;
	if (tiles) *tiles = e264_pred_tiles(h->width_mbs, h->height_mbs);
	return 0;
}

// Everything a kernel will dereference through the packet, checked on the host before the packet may reach the
// device (a wild offset would be a GPU memory fault = process abort, not an error code): section layout, per-macroblock
// kind / slice index / payload bounds / intra modes, reference slots, and the header's summary fields (ref_slots,
// n_coded_mbs, n_inter_mbs) against the records -- the kernels' early exits and the trusted submission path rely on them.
// `slots` (may be null): allocated-slot table of the stream.
// `slot_bytes` (with `slots`): size of every allocated slot -- a packet whose header claims a larger picture than the slot
// it writes or reads (SPS size change, stale capture, foreign packet) would make the kernels run past the allocation.
// `ref_mask_out` (may be null): DPB slots the packet's motion refers to.
struct DeepAcc { uint32_t ref_mask = 0, n_coded = 0, n_inter = 0; bool pred_work = false, has_l1 = false; };
// one macroblock record against its packet: m (the version-4 record), a / col (its address and column), mot / mot_bytes (the motion section its mot_off counts in)
static int check_mb(const E264FrameHdr *h, const E264Mb &m, int a, int col, const uint8_t *mot, uint32_t mot_bytes, uint8_t *const *slots, const size_t *slot_bytes, uint64_t frame_need, DeepAcc &acc)
{
	if (m.kind > E264_MB_INTER) return fail(EINVAL, "macroblock kind");
	if (m.slice >= h->n_slices || m.dbk_slice >= h->n_slices) return fail(EINVAL, "macroblock slice index"); // every record: the parameter kernel reads the slice of absent macroblocks too
	if (m.kind == E264_MB_ABSENT) return 0;
	acc.n_coded++;
	if (m.kind == E264_MB_INTER || m.kind == E264_MB_PCM) acc.pred_work = true; // some macroblock is the prediction kernel's
	if ((m.flags & E264_MBF_T8x8) && (m.kind == E264_MB_I16x16 || m.kind == E264_MB_PCM)) return fail(EINVAL, "8x8 transform flag on an Intra16x16 / PCM macroblock");
	if ((m.payload_off & 7) || (uint64_t)m.payload_off + e264_mb_payload_bytes(&m) > h->payload_bytes) return fail(EINVAL, "macroblock payload");
	if ((m.flags & E264_MBF_EDGE_LEFT) && col == 0) return fail(EINVAL, "left edge flag on the first column");
	if ((m.flags & E264_MBF_EDGE_TOP) && a < h->width_mbs) return fail(EINVAL, "top edge flag on the first row");
	// internal intra modes (src/edge264_internal.h:564-634): the kernels index tables with them
	if (m.kind == E264_MB_I4x4) {
		uint64_t mm;
		memcpy(&mm, m.modes, 8); // 16 nibbles: above 13 <=> bits 1, 2 and 3 all set
		if ((mm >> 1) & (mm >> 2) & (mm >> 3) & 0x1111111111111111ull) return fail(EINVAL, "Intra4x4 mode");
	} else if (m.kind == E264_MB_I8x8) {
		for (int k = 0; k < 4; k++)
			if (m.modes[k] > 31) return fail(EINVAL, "Intra8x8 mode");
	} else if (m.kind == E264_MB_I16x16 && m.i16_mode > 6) return fail(EINVAL, "Intra16x16 mode");
	if (m.kind >= E264_MB_I4x4 && m.kind <= E264_MB_I16x16 && m.chroma_mode > 6) return fail(EINVAL, "intra chroma mode");
	if (m.kind == E264_MB_INTER) {
		acc.n_inter++;
		if (!mot) return fail(EINVAL, "inter macroblock without motion section");
		uint32_t d[2];
		memcpy(d, m.modes, 8); // motion directory: record offset, shape
		if (E264_MOT_UNI(d[1], 1) || (d[1] >> 4 & 15u)) acc.has_l1 = true; // predicts from list 1 (its uniform bit or one of its quadrant bits)
		if ((d[0] & 3) || d[1] >> 26 || (uint64_t)d[0] + e264_mot_record_bytes(d[1]) > mot_bytes) return fail(EINVAL, "macroblock motion record");
		// the record's reference dwords, where they lie (what e264_motion_expand would spread over 8 parts: the uniform form repeats
		// one dword, an unused quadrant of a partitioned list reads as -1, which is always admissible)
		const uint8_t *rec = mot + d[0];
		uint32_t n = 0;
		for (int l = 0; l < 2; l++) {
			const bool uni = E264_MOT_UNI(d[1], l);
			for (int q = 0; q < (uni ? 1 : 4); q++) {
				if (!uni && !E264_MOT_USED(d[1], l * 4 + q)) continue;
				const int rp = (int8_t)rec[n], ri = (int8_t)rec[n + 1];
				if (rp < 0 || rp >= E264_MAX_SLOTS) return fail(EINVAL, "reference slot"); // a part the directory announces predicts from a picture
				if (slots && !slots[rp]) return fail(EINVAL, "reference slot not allocated");
				if (slots && slot_bytes && frame_need > slot_bytes[rp]) return fail(EINVAL, "picture larger than a reference slot");
				acc.ref_mask |= 1u << rp;
				if (ri < -1 || ri > 31) return fail(EINVAL, "reference index");
				n += uni ? 8 : 4 + 4 * e264_mot_nmv(E264_MOT_SUB(d[1], l * 4 + q));
			}
		}
	}
	return 0;
}

static int check_packet_deep(const void *packet, size_t bytes, uint8_t *const *slots, const size_t *slot_bytes = nullptr, uint32_t *ref_mask_out = nullptr, bool *pred_work_out = nullptr, bool *has_l1_out = nullptr)
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
	const uint8_t *mot = h->motion_off ? p + h->motion_off : nullptr; // compact motion records, up to payload_off
	const uint32_t mot_bytes = h->motion_off ? h->payload_off - h->motion_off : 0;
	DeepAcc acc;
	if (h->version == E264_VERSION_COMPACT) {
		// A wire packet (include/edge264_compact.h; check_packet has vetted its structure) means what its expansion means: every entry is held
		// against the same checks as the version-4 record e264_expand_kernel will make of it -- walked in place, entry by entry (the sequential
		// form of e264_expand_mb: running counts instead of popcounts), without unfolding the packet.
		const E264CompactHdr *ch = (const E264CompactHdr *)(p + h->mbs_off);
		const uint32_t wm = h->width_mbs, hm = h->height_mbs, wpr = ch->words_per_row;
		const uint32_t *cbits = (const uint32_t *)(p + h->mbs_off + 16) + 3 * hm, *bbits = cbits + (size_t)hm * wpr;
		const uint8_t *e = p + h->mbs_off + e264_compact_table_bytes(wm, hm);
		for (uint32_t y = 0, a = 0; y < hm; y++)
			for (uint32_t x = 0; x < wm; x++, a++) {
				E264Mb m;
				if (!(cbits[y * wpr + (x >> 5)] >> (x & 31) & 1u)) {
					memcpy(&m, e, 32);
					e += 32;
					if ((r = check_mb(h, m, (int)a, (int)x, mot, mot_bytes, slots, slot_bytes, frame_need, acc))) return r;
					continue;
				}
				const bool both = bbits[y * wpr + (x >> 5)] >> (x & 31) & 1u;
				E264MbCompact k;
				memcpy(&k, e, 12);
				uint8_t rec[16] = {k.ref_slot, k.ref_idx, 0, 0};
				memcpy(rec + 4, k.mv, 4);
				if (both) memcpy(rec + 8, e + 12, 8);
				e += both ? 20 : 12;
				memset(&m, 0, sizeof(m));
				m.kind = E264_MB_INTER; m.flags = (uint8_t)(k.flags & ~E264_MBCF_LIST1);
				m.qp[0] = k.qp[0]; m.qp[1] = k.qp[1]; m.qp[2] = k.qp[2];
				m.slice = k.slice; m.dbk_slice = k.dbk_slice;
				const uint32_t d[2] = {0, both ? E264_MOT_HDR_UNI01 : (k.flags & E264_MBCF_LIST1) ? E264_MOT_HDR_UNI1 : E264_MOT_HDR_UNI0};
				memcpy(m.modes, d, 8);
				if ((r = check_mb(h, m, (int)a, (int)x, rec, both ? 16 : 8, slots, slot_bytes, frame_need, acc))) return r;
			}
	} else {
		const E264Mb *mbs = (const E264Mb *)(p + h->mbs_off);
		for (int a = 0, col = 0; a < n_mbs; a++, col = col + 1 == h->width_mbs ? 0 : col + 1)
			if ((r = check_mb(h, mbs[a], a, col, mot, mot_bytes, slots, slot_bytes, frame_need, acc))) return r;
	}
	if (h->n_coded_mbs != acc.n_coded || h->n_inter_mbs != acc.n_inter) return fail(EINVAL, "header macroblock counts differ from the records");
	if (h->ref_slots != acc.ref_mask) return fail(EINVAL, "header ref_slots differs from the motion records");
	if (ref_mask_out) *ref_mask_out = acc.ref_mask;
	if (pred_work_out) *pred_work_out = acc.pred_work;
	if (has_l1_out) *has_l1_out = acc.has_l1;
	return 0;
}

// host-only entry point of the same checks (tests, front ends that want to vet a capture file)
API int e264hip_packet_check(const void *packet, size_t bytes)
{
	return check_packet_deep(packet, bytes, nullptr);
}

// The wire form (include/edge264_compact.h) for callers that do not compile C: the Python tools, a binding in another language.
API size_t e264hip_packet_compact_bound(const void *packet, size_t bytes)
{
	const E264FrameHdr *h = (const E264FrameHdr *)packet;
	return (packet && bytes >= sizeof(*h) && h->magic == E264_MAGIC && h->version == E264_VERSION) ? e264_compact_bound(packet) : 0;
}
// version 4 -> version 5; the input must pass e264hip_packet_check (checked here).  Returns the size written, 0 on error (e264hip_last_error).
API size_t e264hip_packet_compact(const void *packet, size_t bytes, void *out, size_t cap)
{
	const E264FrameHdr *h = (const E264FrameHdr *)packet;
	if (!packet || !out || bytes < sizeof(*h) || h->version != E264_VERSION || check_packet_deep(packet, bytes, nullptr)) { fail(EINVAL, "packet_compact: not a sound version-4 packet"); return 0; }
	const size_t r = e264_compact_packet(packet, h->total_bytes, out, cap);
	if (!r) fail(EINVAL, "packet_compact: output buffer too small");
	return r;
}
// version 5 -> the canonical version-4 packet.  out == NULL: the size needed.  0 on error.
API size_t e264hip_packet_expand(const void *packet, size_t bytes, void *out, size_t cap)
{
	int dst, n_mbs;
	const E264FrameHdr *h = (const E264FrameHdr *)packet;
	if (check_packet(packet, bytes, &dst, &n_mbs) || h->version != E264_VERSION_COMPACT) { fail(EINVAL, "packet_expand: not a sound wire packet"); return 0; }
	if (!out) return e264_expanded_bytes(packet);
	const size_t r = e264_expand_packet(packet, h->total_bytes, out, cap);
	if (!r) fail(EINVAL, "packet_expand: output buffer too small");
	return r;
}

// A packet that has passed e264hip_packet_check (its header summarises its records) against THIS stream's allocations:
// the destination and every slot of hdr.ref_slots exist and hold a picture of the packet's size.
static int check_slots_of(const E264Stream *s, const E264FrameHdr *h)
{
	const uint64_t need = (uint64_t)h->plane_size_Y + h->plane_size_C;
	if (!s->h_table[h->dst_slot]) return fail(EINVAL, "destination slot not allocated");
	if (need > s->slot_bytes[h->dst_slot]) return fail(EINVAL, "picture larger than the destination slot");
	for (int sl = 0; sl < E264_MAX_SLOTS; sl++)
		if (h->ref_slots >> sl & 1) {
			if (!s->h_table[sl]) return fail(EINVAL, "reference slot not allocated");
			if (need > s->slot_bytes[sl]) return fail(EINVAL, "picture larger than a reference slot");
		}
	return 0;
}

static int ensure_dbk(E264Stream *s, int n_mbs)
{
	if (s->dbk_mbs >= (size_t)n_mbs) return 0;
	mem_release(s->dev, s->d_dbk, E264_SCRATCH_BYTES(s->dbk_mbs), false, mark_lane(s->dev, s->lane), s->lane); // queued kernels may still use it (behind the lane's tail, as frame_free)
	s->d_dbk = (uint8_t *)mem_acquire(s->dev, E264_SCRATCH_BYTES(n_mbs), false);
	s->dbk_mbs = s->d_dbk ? (size_t)n_mbs : 0;
	return s->d_dbk ? 0 : fail(ENOMEM, "hipMalloc deblock parameters");
}

// the stream's expansion buffer (wire packets only)
static int ensure_expand(E264Stream *s, size_t area)
{
	if (!area || s->expand_cap >= area) return 0;
	mem_release(s->dev, s->d_expand, s->expand_cap, false, mark_lane(s->dev, s->lane), s->lane);
	const size_t cap = (area + area / 4 + 65535) & ~(size_t)65535; // the motion section varies from picture to picture
	s->d_expand = (uint8_t *)mem_acquire(s->dev, cap, false);
	s->expand_cap = s->d_expand ? cap : 0;
	return s->d_expand ? 0 : fail(ENOMEM, "hipMalloc expansion buffer");
}

// Launches the kernels over a job table that already lives in HBM, on compute lane `lane`.
// n_nopred: the table's LAST n_nopred jobs hold no inter / PCM macroblock (0: unknown or none)
static int launch(E264Device *dev, int lane, const E264Job *d_jobs, int n, int max_mbs, int max_tiles, int mode, uint64_t *serial_out = nullptr, int n_nopred = 0)
{
	std::lock_guard<std::mutex> g(dev->lock);
	hipEvent_t *marks = nullptr;
	// (not with more than two lanes in use: lanes and second queues then share the runtime's hardware queues -- four by default, GPU_MAX_HW_QUEUES -- and a lane's
	// kernels wait behind another lane's 2.7-ms intra pass: 38.7 k against 52.0 k frames/s without the split, tools/stagger_probe.py, profiles/r06_ablations.txt item 16)
	const bool split = (dev->split_intra == 2 || (dev->split_intra && dev->max_lane.load(std::memory_order_relaxed) < 2)) && dev->q2[lane] && n_nopred > 0 && n_nopred < n;
	// Two workgroups per picture (luma, chroma: e264_intra_planes_kernel) for the pictures whose intra pass stands alone -- while they are few: each takes a whole CU
	// (125 KB of LDS) from the prediction kernel of the others, and with more than ~320 other pictures in the submission their kernels outlast a one-workgroup pass
	// anyway (tools/stagger_probe.py: 256 pictures out of phase 72.4 -> 84.9 k frames/s; 512: 88.7 -> 86.9 k without this rule; profiles/r06_ablations.txt item 18)
	const int cu_budget = dev->n_cus * 3 / 8;
	int planes = 0;
	if (dev->split_planes && split && 2 * n_nopred <= cu_budget && n - n_nopred <= 320) planes |= 1;
	// ... a kernel that has the lane to itself (an all-intra batch's intra pass, every batch's deblocking) may take every CU: 64 streams 38.2 -> 42.8 k frames/s,
	// 128 streams 61.6 -> 69.8 k (gpurun_out/pa1; E264_PLANES_ALONE overrides for A/B)
	static const int alone_env = getenv("E264_PLANES_ALONE") ? atoi(getenv("E264_PLANES_ALONE")) : 0;
	const int cu_alone = alone_env > 0 ? alone_env : dev->n_cus / (dev->max_lane.load(std::memory_order_relaxed) + 1); // (the lanes in use run beside each other: a lane's share)
	if (dev->split_planes && (mode & E264_RUN_NO_PRED) && 2 * n <= cu_alone) planes |= 2;
	if (dev->split_planes && 2 * n <= cu_alone) planes |= 4; // ... and the deblocking kernel's luma and chroma groups (one stream: a P picture 0.89 ms, of which that kernel is most)
	E264Fork fork = {dev->side_queue || split || planes ? dev->q2[lane] : nullptr, dev->forked[lane], dev->joined[lane], nullptr, dev->side_queue, split ? n_nopred : 0, planes};
	if (dev->ktiming) {
		if (dev->kev_used == dev->kev.size()) {
			E264Device::Marks m;
			for (int i = 0; i < 5; i++) hipEventCreate(&m.e[i]);
			for (int i = 0; i < 2; i++) hipEventCreate(&m.a[i]);
			dev->kev.push_back(m);
		}
		E264Device::Marks &m = dev->kev[dev->kev_used++];
		m.side = !split && dev->side_queue && fork.aux != nullptr && (mode & 2); // (the parameter kernel's own marks; in a split submission it stays on the lane and the intra phase [2..3] holds the join)
		marks = m.e; fork.amarks = m.a;
	}
	HIPCHK(e264_launch_frames(d_jobs, n, max_mbs, max_tiles, mode, dev->waves | dev->intra_waves << 8, dev->q[lane], marks, &fork), EIO);
	const uint64_t serial = ++dev->serial;
	const int idx = (int)(serial % E264Device::NEV);
	if (dev->sub_ev[idx] && hipEventRecord(dev->sub_ev[idx], dev->q[lane]) == hipSuccess) { dev->sub_serial[idx] = serial; dev->sub_lane[idx] = lane; }
	else dev->sub_serial[idx] = 0;
	if (serial_out) *serial_out = serial;
	return 0;
}

// A serial for work queued on a lane OUTSIDE a batch submission (a fill): same numbering, same event ring, so that whatever
// waits for "the stream's latest work" or parks memory behind it covers the fill too.
static uint64_t mark_lane(E264Device *dev, int lane)
{
	std::lock_guard<std::mutex> g(dev->lock);
	const uint64_t serial = ++dev->serial;
	const int idx = (int)(serial % E264Device::NEV);
	if (dev->sub_ev[idx] && hipEventRecord(dev->sub_ev[idx], dev->q[lane]) == hipSuccess) { dev->sub_serial[idx] = serial; dev->sub_lane[idx] = lane; }
	else dev->sub_serial[idx] = 0;
	return serial;
}

// Blocks until the submission that wrote `slot` last has retired (not until the device is idle).
static int wait_slot(E264Stream *s, int slot)
{
	if (!s->slot_serial[slot]) { // filled / uploaded outside a submission: the stream's lane
		HIPCHK(hipStreamSynchronize(lane_of(s)), EIO);
		return 0;
	}
	return serial_wait(s->dev, s->slot_serial[slot], s->lane);
}

// The next slot of the stream's staging ring, large enough for max_bytes: pinned host + device buffer + job slot + event.
static E264Stream::Stage *stage_prepare(E264Stream *s, size_t max_bytes)
{
	E264Stream::Stage &st = s->stage[s->stage_next];
	if (st.busy) { hipEventSynchronize(st.done); st.busy = false; }
	if (!st.done && hipEventCreateWithFlags(&st.done, E264_WAIT_EVENT) != hipSuccess) { st.done = nullptr; fail(EIO, "hipEventCreate"); return nullptr; }
	if (!st.d_job && !(st.d_job = (E264Job *)mem_acquire(s->dev, sizeof(E264Job), false))) { fail(ENOMEM, "job slot"); return nullptr; }
	if (st.cap < max_bytes) {
		mem_release(s->dev, st.h, st.cap + 64, true, 0, 0); // not busy: nothing in flight reads it
		mem_release(s->dev, st.d, st.cap, false, 0, 0);
		st.h = nullptr; st.d = nullptr; st.cap = 0;
		size_t cap = (max_bytes + 65535) & ~(size_t)65535;
		if (!(st.h = mem_acquire(s->dev, cap + 64, true))) { fail(ENOMEM, "pinned packet buffer"); return nullptr; }
		if (!(st.d = (uint8_t *)mem_acquire(s->dev, cap, false))) { mem_release(s->dev, st.h, cap + 64, true, 0, 0); st.h = nullptr; fail(ENOMEM, "device packet buffer"); return nullptr; }
		st.cap = cap;
	}
	return &st;
}

API void *e264hip_packet_buffer(E264Stream *s, size_t max_bytes)
{
	if (!s || set_device(s->dev)) return nullptr;
	E264Stream::Stage *st = stage_prepare(s, max_bytes);
	return st ? st->h : nullptr;
}

API int e264hip_frame_submit(E264Stream *s, const void *packet, size_t bytes)
{
	if (!s) return fail(EINVAL, "null stream");
	size_t area = 0;
	int dst, n_mbs, n_tiles, r = check_packet(packet, bytes, &dst, &n_mbs, &n_tiles, &area);
	if (r) return r;
	if (!s->h_table[dst]) return fail(EINVAL, "destination slot not allocated");
	bool has_l1 = true, pred_work = true;
	if ((r = check_packet_deep(packet, bytes, s->h_table, s->slot_bytes, nullptr, &pred_work, &has_l1))) return r;
	if (set_device(s->dev)) return EIO;
	if ((r = ensure_dbk(s, n_mbs)) || (r = ensure_expand(s, area))) return r;
	E264Stream::Stage *st = &s->stage[s->stage_next];
	if (packet == st->h && bytes > st->cap) return fail(EINVAL, "packet larger than the buffer e264hip_packet_buffer returned");
	if (packet != st->h) { // caller did not use our pinned buffer: stage it
		st = stage_prepare(s, bytes);
		if (!st) return ENOMEM;
		memcpy(st->h, packet, bytes);
	}
	s->stage_next = (s->stage_next + 1) & 3;
	hipStream_t q = lane_of(s);
	// the job record rides at the tail of the pinned staging buffer: tiny H2D on the same queue
	E264Job *job = (E264Job *)((uint8_t *)st->h + st->cap); // pinned, lives as long as the staging slot
	job->packet = st->d; job->dpb = s->d_table; job->dbk = s->d_dbk; job->expand = area ? s->d_expand : nullptr;
	hipError_t e = hipMemcpyAsync(st->d, st->h, bytes, hipMemcpyHostToDevice, q);
	if (e == hipSuccess) e = hipMemcpyAsync(st->d_job, job, sizeof(*job), hipMemcpyHostToDevice, q);
	uint64_t serial = 0;
	r = e == hipSuccess ? launch(s->dev, s->lane, st->d_job, 1, n_mbs, n_tiles, E264_RUN_ALL | (has_l1 ? 0 : E264_RUN_NO_L1) | (pred_work ? 0 : E264_RUN_NO_PRED) | (area ? E264_RUN_EXPAND : 0), &serial) : fail(EIO, "hipMemcpyAsync packet", e);
	// whatever was queued reads the staging slot: it is busy until the lane has passed this point, error or not
	hipEventRecord(st->done, q);
	st->busy = true;
	if (r) return r;
	raise_serial(s->slot_serial[dst], serial);
	raise_serial(s->last_serial, serial);
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
	hipStream_t qc = s->dev->qc ? s->dev->qc : lane_of(s);
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
	bool pred_work = true, has_l1 = true;
	if ((r = check_packet_deep(packet, bytes, nullptr, nullptr, &ref_mask, &pred_work, &has_l1))) return r;
	std::vector<uint8_t> unfolded; // a packet that stays in HBM is kept as the kernels read it: a wire packet is unfolded once, here
	if (((const E264FrameHdr *)packet)->version == E264_VERSION_COMPACT) {
		unfolded.resize(e264_expanded_bytes(packet));
		bytes = e264_expand_packet(packet, ((const E264FrameHdr *)packet)->total_bytes, unfolded.data(), unfolded.size());
		packet = unfolded.data();
	}
	if (set_device(dev)) return EIO;
	E264Packet *p = new (std::nothrow) E264Packet();
	if (!p) return fail(ENOMEM, "packet object");
	p->dev = dev; p->bytes = bytes; p->dst_slot = dst; p->n_mbs = n_mbs; p->n_tiles = n_tiles; p->ref_mask = ref_mask; p->pred_work = pred_work; p->has_l1 = has_l1;
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
	e264hip_device_sync(p->dev); // a resident packet may be named by batches of any lane (benchmarks, tests: not a decoder's path)
	hipFree(p->d_bytes);
	delete p;
}

struct E264Batch {
	E264Device *dev;
	E264Job *d_jobs;
	int n, max_mbs, max_tiles, lane;
	int n_nopred;   // the table's last n_nopred jobs are pictures without inter / PCM macroblocks
	bool pred_work; // some packet of the batch has inter / PCM macroblocks (all-intra batches skip the prediction kernel's launch)
	bool has_l1;    // some packet of the batch predicts from list 1
	std::vector<std::pair<E264Stream *, int>> writes; // (stream, destination slot) of every job
};

API int e264hip_batch_create(E264Device *dev, E264Stream *const *streams, E264Packet *const *packets, int n, E264Batch **out)
{
	if (!dev || !streams || !packets || !out || n <= 0) return fail(EINVAL, "batch_create arguments");
	if (set_device(dev)) return EIO;
	std::vector<E264Job> jobs((size_t)n);
	int max_mbs = 0, max_tiles = 0;
	size_t n_front = 0, n_back = 0;
	for (int i = 0; i < n; i++) {
		if (!streams[i] || !packets[i] || streams[i]->dev != dev || packets[i]->dev != dev) return fail(EINVAL, "batch entry");
		if (streams[i]->lane != streams[0]->lane) return fail(EINVAL, "the streams of a batch must be bound to one compute lane");
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
		// the table's order is the launcher's to choose: pictures without prediction work (I pictures) LAST, so that a mixed batch can start their intra pass
		// beside the others' parameter and prediction kernels (E264Fork.n_nopred)
		E264Job &jb = jobs[packets[i]->pred_work ? n_front++ : (size_t)n - 1 - n_back++];
		jb.packet = packets[i]->d_bytes;
		jb.dpb = streams[i]->d_table;
		jb.dbk = streams[i]->d_dbk;
		jb.expand = nullptr; // (resident packets are version 4: e264hip_packet_upload)
		if (packets[i]->n_mbs > max_mbs) max_mbs = packets[i]->n_mbs;
		if (packets[i]->n_tiles > max_tiles) max_tiles = packets[i]->n_tiles;
	}
	E264Batch *b = new (std::nothrow) E264Batch();
	if (!b) return fail(ENOMEM, "batch object");
	b->dev = dev; b->n = n; b->max_mbs = max_mbs; b->max_tiles = max_tiles; b->lane = streams[0]->lane; b->n_nopred = (int)n_back;
	b->pred_work = b->has_l1 = false;
	for (int i = 0; i < n; i++) { b->pred_work = b->pred_work || packets[i]->pred_work; b->has_l1 = b->has_l1 || packets[i]->has_l1; }
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
	for (auto &w : b->writes)
		if (w.first->lane != b->lane) return fail(EINVAL, "a stream of the batch was bound to another lane after batch_create");
	uint64_t serial = 0;
	// (E264_RUN_NO_PRED: internal to the launcher -- every packet of the batch was vetted at upload time and none holds an inter or PCM macroblock)
	int r = launch(b->dev, b->lane, b->d_jobs, b->n, b->max_mbs, b->max_tiles, (mode & E264_RUN_ALL) | (b->pred_work ? 0 : E264_RUN_NO_PRED) | (b->has_l1 ? 0 : E264_RUN_NO_L1), &serial, b->n_nopred);
	if (!r) for (auto &w : b->writes) { raise_serial(w.first->slot_serial[w.second], serial); raise_serial(w.first->last_serial, serial); }
	return r;
}

API void e264hip_batch_free(E264Batch *b)
{
	if (!b) return;
	hipSetDevice(b->dev->ordinal);
	hipStreamSynchronize(b->dev->q[b->lane]);
	hipFree(b->d_jobs);
	delete b;
}

API int e264hip_submit_batch(E264Device *dev, E264Stream *const *streams, E264Packet *const *packets, int n, int mode)
{
	E264Batch *b = nullptr;
	int r = e264hip_batch_create(dev, streams, packets, n, &b);
	if (r) return r;
	r = e264hip_batch_submit(b, mode);
	e264hip_batch_free(b); // synchronises the lane: convenience path for tests
	return r;
}

// Host packets of MANY streams, one submission, nothing synchronous: every packet is staged through its
// stream's pinned ring and copied on the upload queue, the job table through a device-level ring, then the four
// kernels are launched on the streams' lane behind the batch's upload event.  The caller may reuse / free the host
// packets on return.
static int submit_host_impl(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode, int flags);
API int e264hip_submit_batch_host(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode)
{
	return submit_host_impl(dev, streams, packets, bytes, n, mode, 0);
}
// The same for packets that already sit in page-locked memory (e264hip_host_alloc; what a front end does when it assembles
// the packet in place): no staging copy, the H2D reads the caller's buffer, which must stay untouched until the submission
// has retired (e264hip_device_sync / a later e264hip_frame_wait).  E264_SUBMIT_TRUSTED: the caller has run
// e264hip_packet_check on exactly these bytes (a packet produced by its own emitter, a validated capture): the
// per-macroblock walk is not repeated on the submitting thread; the slots the vetted header names (dst_slot, ref_slots)
// are still checked against this stream's allocations.
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
// What the launcher wants to know about a packet its producer has vetted (E264_SUBMIT_TRUSTED: the header summarises the records, the records are sound) without the
// per-macroblock walk: does any macroblock have work for the prediction kernel (inter, I_PCM), does any predict from list 1 (else the parameter kernel's small form
// will do)?  One byte / one dword per record: ~10 us per 1080p packet, on the thread that gathers it.
static void scan_trusted(const void *packet, int n_mbs, bool *pred_work, bool *has_l1)
{
	const E264FrameHdr *h = (const E264FrameHdr *)packet;
	const uint8_t *p = (const uint8_t *)packet;
	*pred_work = true; *has_l1 = true;
	if (h->version == E264_VERSION) {
		const uint8_t *rec = p + h->mbs_off;
		bool pw = h->n_inter_mbs != 0, l1 = false;
		for (int a = 0; a < n_mbs && !(pw && l1); a++, rec += sizeof(E264Mb)) {
			if (rec[0] == E264_MB_PCM) pw = true;
			else if (rec[0] == E264_MB_INTER) {
				uint32_t mh;
				memcpy(&mh, rec + offsetof(E264Mb, modes) + 4, 4);
				if (E264_MOT_UNI(mh, 1) || (mh >> 4 & 15u)) l1 = true;
			}
			if (!h->n_inter_mbs && pw) break; // (no inter macroblock: nothing more to learn)
		}
		*pred_work = pw; *has_l1 = l1;
	} else if (h->version == E264_VERSION_COMPACT) { // (folded: it has inter macroblocks; its structure was checked by check_packet)
		const E264CompactHdr *ch = (const E264CompactHdr *)(p + h->mbs_off);
		if (ch->n_both) return;
		const uint32_t wm = h->width_mbs, hm = h->height_mbs, wpr = ch->words_per_row;
		const uint32_t *cbits = (const uint32_t *)(p + h->mbs_off + 16) + 3 * hm;
		const uint8_t *e = p + h->mbs_off + e264_compact_table_bytes(wm, hm);
		bool l1 = false;
		for (uint32_t y = 0; y < hm && !l1; y++)
			for (uint32_t x = 0; x < wm && !l1; x++) {
				if (cbits[y * wpr + (x >> 5)] >> (x & 31) & 1u) { l1 = e[0] & E264_MBCF_LIST1; e += 12; continue; } // (no two-list entry: n_both == 0)
				if (e[0] == E264_MB_INTER) {
					uint32_t mh;
					memcpy(&mh, e + offsetof(E264Mb, modes) + 4, 4);
					l1 = E264_MOT_UNI(mh, 1) || (mh >> 4 & 15u);
				}
				e += 32;
			}
		*has_l1 = l1;
	}
}

static int gather_workers()
{
	static const int v = [] { const char *e = getenv("E264_GATHER_THREADS"); return e ? atoi(e) : E264_GATHER_WORKERS; }();
	return v;
}
static int submit_host_impl(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode, int flags)
{
	const bool pinned = flags & 1, trusted = flags & 2;
	// Page-locked packets of a LARGE batch are gathered into the batch's staging buffer like pageable ones (host threads, ~10 GB/s each) and
	// cross PCIe as one transfer: 256 separate 1-MB copies reach 35 GB/s (36 k frames/s on the bench GOP), one 246-MB copy 54 k.
	const bool stage = !pinned || n >= 32;
	if (!dev || !streams || !packets || !bytes || n <= 0) return fail(EINVAL, "submit_batch_host arguments");
	if (set_device(dev)) return EIO;
	std::vector<int> mbs_of((size_t)n), tiles_of((size_t)n), rc((size_t)n, 0), dst_of((size_t)n);
	std::vector<char> l1_of((size_t)n, 1), pw_of((size_t)n, 1); // (trusted packets are not walked here: scan_trusted)
	std::vector<std::string> why((size_t)n);
	for (int i = 0; i < n; i++) {
		if (!streams[i] || streams[i]->dev != dev) return fail(EINVAL, "batch entry");
		if (streams[i]->lane != streams[0]->lane) return fail(EINVAL, "the streams of a batch must be bound to one compute lane");
		for (int j = 0; j < i; j++)
			if (streams[j] == streams[i]) return fail(EINVAL, "a stream may contribute one frame per batch");
	}
	const int lane = streams[0]->lane;
	// headers first (cheap, serial): sizes, destination slots
	int max_mbs = 0, max_tiles = 0;
	std::vector<size_t> off_of((size_t)n), area_of((size_t)n), xoff_of((size_t)n);
	size_t total = 0, xtotal = 0;
	bool any_wire = false;
	for (int i = 0; i < n; i++) {
		int r = check_packet(packets[i], bytes[i], &dst_of[i], &mbs_of[i], &tiles_of[i], &area_of[i]);
		if (r) return r;
		any_wire = any_wire || area_of[i];
		xoff_of[i] = xtotal;
		xtotal += (area_of[i] + 255) & ~(size_t)255;
		if (!streams[i]->h_table[dst_of[i]]) return fail(EINVAL, "destination slot not allocated");
		if (tiles_of[i] > max_tiles) max_tiles = tiles_of[i];
		if (mbs_of[i] > max_mbs) max_mbs = mbs_of[i];
		off_of[i] = total;
		total += (bytes[i] + 255) & ~(size_t)255;
	}
	std::lock_guard<std::mutex> bg(dev->batch_lock); // batches of one device are serialised (their streams are disjoint per batch anyway)
	// ---- every allocation first: a failure below this block would leave copies in flight on buffers nobody guards ----
	E264Device::JobRing &jr = dev->jring[dev->jring_next];
	if (jr.busy) { hipEventSynchronize(jr.done); jr.busy = false; }
	if (jr.cap < n) {
		if (jr.h) mem_release(dev, jr.h, sizeof(E264Job) * jr.cap, true, 0, 0);
		if (jr.d) mem_release(dev, jr.d, sizeof(E264Job) * jr.cap, false, 0, 0);
		jr.h = nullptr; jr.d = nullptr; jr.cap = 0;
		int cap = (n + 63) & ~63;
		if (!(jr.h = (E264Job *)mem_acquire(dev, sizeof(E264Job) * cap, true))) return fail(ENOMEM, "pinned job table");
		if (!(jr.d = (E264Job *)mem_acquire(dev, sizeof(E264Job) * cap, false))) { mem_release(dev, jr.h, sizeof(E264Job) * cap, true, 0, 0); jr.h = nullptr; return fail(ENOMEM, "device job table"); }
		jr.cap = cap;
	}
	if (stage && jr.pcap < total) { // the batch's staging: a quarter of headroom, pictures of a stream differ in size
		if (jr.ph) mem_release(dev, jr.ph, jr.pcap, true, 0, 0);
		if (jr.pd) mem_release(dev, jr.pd, jr.pcap, false, 0, 0);
		jr.ph = jr.pd = nullptr; jr.pcap = 0;
		const size_t cap = (total + total / 4 + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
		if (!(jr.ph = (uint8_t *)mem_acquire(dev, cap, true))) return fail(ENOMEM, "pinned batch staging");
		if (!(jr.pd = (uint8_t *)mem_acquire(dev, cap, false))) { mem_release(dev, jr.ph, cap, true, 0, 0); jr.ph = nullptr; return fail(ENOMEM, "device batch staging"); }
		jr.pcap = cap;
	}
	if (jr.xcap < xtotal) {
		if (jr.xd) mem_release(dev, jr.xd, jr.xcap, false, 0, 0); // not busy: nothing in flight reads it
		jr.xd = nullptr; jr.xcap = 0;
		const size_t cap = (xtotal + xtotal / 4 + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
		if (!(jr.xd = (uint8_t *)mem_acquire(dev, cap, false))) return fail(ENOMEM, "device expansion buffer");
		jr.xcap = cap;
	}
	if (!jr.done && hipEventCreateWithFlags(&jr.done, E264_WAIT_EVENT) != hipSuccess) { jr.done = nullptr; return fail(EIO, "hipEventCreate"); }
	if (!jr.up && hipEventCreateWithFlags(&jr.up, hipEventDisableTiming) != hipSuccess) { jr.up = nullptr; return fail(EIO, "hipEventCreate"); }
	std::vector<E264Stream::Stage *> stage_of((size_t)n, nullptr);
	for (int i = 0; i < n; i++) { // per-stream buffers (HIP calls: this thread only); the rings advance only when everything is there
		E264Stream *s = streams[i];
		int r = ensure_dbk(s, mbs_of[i]);
		if (r) return r;
		if (!stage && !(stage_of[i] = stage_prepare(s, bytes[i]))) return ENOMEM;
	}
	{ // every packet on its own, in parallel: the per-macroblock walk, and -- while its lines are still in the core's cache -- the copy
	  // into the batch's staging buffer.  Nothing has been queued yet: a packet that fails leaves no trace (the ring has not advanced).
		std::lock_guard<std::mutex> pg(dev->pool_user);
		dev->pool.parallel_for(n, [&](int i) {
			E264Stream *s = streams[i];
			bool l1 = true, pw = true;
			int r = trusted ? check_slots_of(s, (const E264FrameHdr *)packets[i]) : check_packet_deep(packets[i], bytes[i], s->h_table, s->slot_bytes, nullptr, &pw, &l1);
			if (trusted && !r) scan_trusted(packets[i], mbs_of[i], &pw, &l1);
			l1_of[i] = l1; pw_of[i] = pw;
			if (r) { rc[i] = r; why[i] = g_err; return; } // the message lives in the worker's thread-local buffer
			if (stage) memcpy(jr.ph + off_of[i], packets[i], bytes[i]);
		}, trusted ? gather_workers() : 0);
	}
	for (int i = 0; i < n; i++)
		if (rc[i]) return fail(rc[i], why[i].c_str());
	dev->jring_next = (dev->jring_next + 1) % E264_JOB_RING;
	if (!stage) for (int i = 0; i < n; i++) streams[i]->stage_next = (streams[i]->stage_next + 1) & 3;
	// ---- copies on the upload queue, kernels on the lane behind the batch's upload event ----
	hipStream_t q = dev->q[lane], up = dev->upload_queue && dev->qup ? dev->qup : q;
	hipError_t e = hipSuccess;
	int n_front = 0, n_back = 0; // pictures without prediction work last in the table (as in e264hip_batch_create)
	for (int i = 0; i < n; i++) {
		E264Job &jb = jr.h[pw_of[i] ? n_front++ : n - 1 - n_back++];
		jb.packet = stage ? jr.pd + off_of[i] : stage_of[i]->d;
		jb.dpb = streams[i]->d_table; jb.dbk = streams[i]->d_dbk;
		jb.expand = area_of[i] ? jr.xd + xoff_of[i] : nullptr;
	}
	if (!stage)
		for (int i = 0; i < n && e == hipSuccess; i++)
			e = hipMemcpyAsync(stage_of[i]->d, packets[i], bytes[i], hipMemcpyHostToDevice, up);
	else
		e = hipMemcpyAsync(jr.pd, jr.ph, total, hipMemcpyHostToDevice, up);
	if (e == hipSuccess) e = hipMemcpyAsync(jr.d, jr.h, sizeof(E264Job) * n, hipMemcpyHostToDevice, up);
	if (e == hipSuccess && any_wire) e = e264_launch_expand(jr.d, n, max_mbs, up);
	if (e == hipSuccess && up != q) {
		e = hipEventRecord(jr.up, up);
		if (e == hipSuccess) e = hipStreamWaitEvent(q, jr.up, 0);
	}
	bool batch_l1 = false;
	for (int i = 0; i < n; i++) batch_l1 = batch_l1 || l1_of[i];
	uint64_t serial = 0;
	int r = e == hipSuccess ? launch(dev, lane, jr.d, n, max_mbs, max_tiles, (mode & E264_RUN_ALL) | (batch_l1 ? 0 : E264_RUN_NO_L1) | (n_back == n ? E264_RUN_NO_PRED : 0), &serial, n_back) : fail(EIO, "packet upload", e);
	if (r && up != q) hipStreamSynchronize(up); // copies already queued must not outlive the error return unguarded
	// the job table and the staging slots are busy until the lane has passed this point -- also on an error above: whatever
	// part of the batch was queued still reads them
	hipEventRecord(jr.done, q);
	jr.busy = true;
	if (!stage)
		for (int i = 0; i < n; i++) {
			hipEventRecord(stage_of[i]->done, q);
			stage_of[i]->busy = true;
		}
	if (r) return r;
	for (int i = 0; i < n; i++) { raise_serial(streams[i]->slot_serial[dst_of[i]], serial); raise_serial(streams[i]->last_serial, serial); }
	return 0;
}

// Timing events are recorded on lane 0 AFTER everything queued on the other lanes so far (lane 0 waits for them): with one
// lane in use that is the plain event on the queue of the kernels, with several it brackets all of them.
API int e264hip_event_record(E264Device *dev, int idx)
{
	if (!dev || idx < 0 || idx >= 16) return fail(EINVAL, "event index");
	if (set_device(dev)) return EIO;
	for (int i = 1; i < E264Device::NQ; i++) {
		HIPCHK(hipEventRecord(dev->lane_ev[i], dev->q[i]), EIO);
		HIPCHK(hipStreamWaitEvent(dev->q[0], dev->lane_ev[i], 0), EIO);
	}
	HIPCHK(hipEventRecord(dev->ev[idx], dev->q[0]), EIO);
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

// has everything queued before e264hip_event_record(dev, idx) left the GPU?  0 yes, EBUSY not yet (never blocks)
API int e264hip_event_query(E264Device *dev, int idx)
{
	if (!dev || idx < 0 || idx >= 16) return fail(EINVAL, "event index");
	if (set_device(dev)) return EIO;
	const hipError_t e = hipEventQuery(dev->ev[idx]);
	if (e == hipSuccess) return 0;
	if (e == hipErrorNotReady) { (void)hipGetLastError(); return EBUSY; }
	return fail(EIO, "hipEventQuery");
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
