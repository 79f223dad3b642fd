// e264_multi -- batched multi-stream front end (the N-stream counterpart of the
// reference's `edge264_test -b`, test.c:427-546).
//
// N decoder instances (edge264.h API, served by the reference's parsers + our emitters in sink mode 2: frames
// live in HBM, finished command packets are queued).  Host parsing (the reference front end with n_threads = 0) is
// spread over --threads T host threads, each owning a fixed subset of the decoders and parsing up to --ahead K pictures
// ahead of the device; the main thread collects the oldest queued packet of every decoder that has one and sends them
// to the GPU as ONE batch (e264hip_submit_batch_host: 4 kernel launches for all of them; a decoder contributes at most
// one picture per batch, its pictures stay in order).  Parsing and device work overlap; nothing waits for a "round"
// (the first version advanced all decoders by one picture between two barriers: every round took as long as its slowest
// picture -- an I picture with CABAC -- and 64 threads parsed 4.7 k pictures/s where one parses 400).
//
//   e264_multi --front <libedge264_hipfront.so> --hip <libedge264_hip.so> [--device N | --devices 0,1,...] [--repeat R]
//              [--threads T] [--out DIR] [--dump-packets FILE] [--parse-only] [--no-download] a.264 b.264 ...
//
// --devices: decoder k lives on GPU devices[k mod N] (streams are independent: no traffic between GPUs); every
// GPU has its own submitter thread and its own batches.
// --out writes s<k>.yuv (cropped Y, Cb, Cr planes of every output frame, as README.md:126-155 of the reference
// does); --dump-packets appends every command packet (self-describing: E264FrameHdr.total_bytes) = the capture
// format of SURVEY.md 8(f) rank 2.  Prints one JSON line with the throughput.
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// edge264.h:45-62 (layout restated; the header itself is not part of this repository)
struct Edge264Frame {
	const uint8_t *samples[3];
	const uint8_t *samples_mvc[3];
	const uint8_t *mb_errors;
	int8_t bit_depth_Y, bit_depth_C;
	int16_t width_Y, width_C, height_Y, height_C, stride_Y, stride_C, stride_mb;
	int32_t FrameId, FrameId_mvc;
	int16_t frame_crop_offsets[4];
	void *return_arg;
};

struct Front {
	void *(*alloc)(int, void *, void *, int, void *, void *, void *);
	int (*decode_NAL)(void *, const uint8_t *, const uint8_t *, void *, void *);
	int (*get_frame)(void *, Edge264Frame *, int);
	void (*free_dec)(void **);
	void (*flush)(void *);
	const uint8_t *(*find_start_code)(const uint8_t *, const uint8_t *, int);
	void (*set_sink)(int);
	void (*set_device)(int);
	void (*set_download)(int);
	void (*set_pinned)(int);
	int (*take_packet)(void *, void **, size_t *);
	void (*free_packet)(void *);
	void *(*stream)(void *);
	void *(*device)(void);
	void *(*device_of)(void *);
};
struct Hip {
	int (*submit_batch_host)(void *, void **, const void **, const size_t *, int, int);
	int (*submit_batch_pinned)(void *, void **, const void **, const size_t *, int, int, int);
	int (*event_record)(void *, int);
	int (*event_query)(void *, int);
	int (*device_sync)(void *);
	const char *(*last_error)(void);
};

template <typename T> static void bind(void *lib, const char *name, T &fn)
{
	fn = reinterpret_cast<T>(dlsym(lib, name));
	if (!fn) { fprintf(stderr, "e264_multi: missing symbol %s\n", name); exit(2); }
}

struct Stream {
	std::vector<uint8_t> data;
	const uint8_t *nal = nullptr, *end = nullptr;
	void *dec = nullptr;
	bool done = false;
	int loops_left = 0;
	const uint8_t *first_nal = nullptr;
	long frames = 0;
	FILE *out = nullptr;
	struct Pkt { void *data; size_t bytes; };
	std::deque<Pkt> q;     // parsed, not yet submitted (guarded by the queue mutex)
	std::atomic<bool> finished{false}; // its worker will queue nothing more (read outside `mu` by the other workers)
	std::atomic<bool> held{false}; // a parser thread is inside this decoder right now
	int dev_index = 0;     // which of --devices holds its frames
	bool just_flushed = false; // the last thing done to it was edge264_flush, to get it out of ENOBUFS with nothing to fetch
};

static void write_frame(FILE *f, const Edge264Frame &fr)
{
	for (int v = 0; v < 2; v++) { // MVC: the second view follows the base view (edge264.h:46-47)
		const uint8_t *const *pl = v ? fr.samples_mvc : fr.samples;
		if (!pl[0]) continue;
		for (int y = 0; y < fr.height_Y; y++) fwrite(pl[0] + (size_t)y * fr.stride_Y, 1, fr.width_Y, f);
		for (int p = 1; p < 3; p++)
			for (int y = 0; y < fr.height_C; y++) fwrite(pl[p] + (size_t)y * fr.stride_C, 1, fr.width_C, f);
	}
}

static void on_crash(int sig)
{ // a crash inside the (foreign) parser is otherwise silent: print where
	void *bt[48];
	int n = backtrace(bt, 48);
	fprintf(stderr, "e264_multi: signal %d\n", sig);
	backtrace_symbols_fd(bt, n, 2);
	_exit(128 + sig);
}

static bool g_pinned = true;

int main(int argc, char **argv)
{
	signal(SIGSEGV, on_crash);
	signal(SIGBUS, on_crash);
	std::string front_path, hip_path, out_dir, dump_path;
	std::vector<int> devices; // --devices
	int ahead = 3; // --ahead K: pictures a decoder may be parsed ahead of the device
	int device = 0, repeat = 1, n_threads = 1, loops = 1; // --loops K: every stream is played K times back to back (steady state)
	bool no_download = false; // --no-download: output frames stay in HBM (edge264_get_frame does not copy them back)
	bool &pinned = g_pinned; // --pageable turns it off: packets are assembled in page-locked buffers (e264front_set_pinned) and submitted in place
	                    // (e264hip_submit_batch_pinned, E264_SUBMIT_TRUSTED: the producer is the emitter) instead of being validated
	                    // again and copied into staging memory by the back end
	bool pin = false; // --pin: parser thread k runs on the k-th CPU this process may use (the submitters float): no migration, warm caches
	bool stay = false; // --stay: a thread parses up to `ahead` pictures of one decoder in a row (round 5: cache locality against rotation, profiles/r05_host.txt)
	bool parse_only = false; // --parse-only: sink 1, no GPU: packets are produced and dropped (front-end speed / debugging)
	std::vector<std::string> files;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto next = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
		if (a == "--devices") { std::string v = next(); for (size_t p = 0; p < v.size();) { size_t e = v.find(',', p); if (e == std::string::npos) e = v.size(); devices.push_back(atoi(v.substr(p, e - p).c_str())); p = e + 1; } }
		else if (a == "--front") front_path = next();
		else if (a == "--hip") hip_path = next();
		else if (a == "--device") device = atoi(next().c_str());
		else if (a == "--repeat") repeat = atoi(next().c_str());
		else if (a == "--parse-only") parse_only = true;
		else if (a == "--no-download") no_download = true;
		else if (a == "--pageable") pinned = false;
		else if (a == "--threads") n_threads = atoi(next().c_str());
		else if (a == "--loops") loops = atoi(next().c_str());
		else if (a == "--ahead") ahead = std::max(1, atoi(next().c_str()));
		else if (a == "--pin") pin = true;
		else if (a == "--stay") stay = true;
		else if (a == "--out") out_dir = next();
		else if (a == "--dump-packets") dump_path = next();
		else files.push_back(a);
	}
	if (front_path.empty() || hip_path.empty() || files.empty()) {
		fprintf(stderr, "usage: e264_multi --front libedge264_hipfront.so --hip libedge264_hip.so [--device N] [--repeat R] [--out DIR] [--dump-packets FILE] a.264 ...\n");
		return 2;
	}
	setenv("E264_HIP_LIB", hip_path.c_str(), 1); // the front end binds the same back-end library
	void *hl = dlopen(hip_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
	if (!hl && !parse_only) { fprintf(stderr, "e264_multi: %s\n", dlerror()); return 2; } // no back end, no decoding: there is no CPU fallback
	void *fl = dlopen(front_path.c_str(), RTLD_NOW | RTLD_LOCAL);
	if (!fl) { fprintf(stderr, "e264_multi: %s\n", dlerror()); return 2; }
	Front F; Hip H;
	bind(fl, "edge264_alloc", F.alloc); bind(fl, "edge264_decode_NAL", F.decode_NAL); bind(fl, "edge264_get_frame", F.get_frame);
	bind(fl, "edge264_free", F.free_dec); bind(fl, "edge264_flush", F.flush); bind(fl, "edge264_find_start_code", F.find_start_code);
	bind(fl, "e264front_set_sink", F.set_sink); bind(fl, "e264front_set_device", F.set_device); bind(fl, "e264front_set_download", F.set_download);
	bind(fl, "e264front_set_pinned", F.set_pinned);
	bind(fl, "e264front_take_packet", F.take_packet); bind(fl, "e264front_free_packet", F.free_packet);
	bind(fl, "e264front_stream", F.stream); bind(fl, "e264front_device", F.device); bind(fl, "e264front_device_of", F.device_of);
	if (!parse_only) {
	bind(hl, "e264hip_submit_batch_host", H.submit_batch_host); bind(hl, "e264hip_device_sync", H.device_sync);
	bind(hl, "e264hip_submit_batch_pinned", H.submit_batch_pinned); bind(hl, "e264hip_event_record", H.event_record); bind(hl, "e264hip_event_query", H.event_query);
	bind(hl, "e264hip_last_error", H.last_error);
	}

	if (devices.empty()) devices.push_back(device);
	F.set_sink(parse_only ? 1 : 2);
	F.set_download(no_download ? 0 : 1);
	if (parse_only) pinned = false;
	F.set_pinned(pinned ? 1 : 0);
	std::deque<Stream> S; // (a deque never moves its elements: Stream keeps pointers into its own data and is not movable)
	for (int r = 0; r < repeat; r++)
		for (const std::string &path : files) {
			FILE *f = fopen(path.c_str(), "rb");
			if (!f) { perror(path.c_str()); return 2; }
			fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
			S.emplace_back();
			Stream &t = S.back();
			t.data.resize((size_t)n + 64);
			if (fread(t.data.data(), 1, (size_t)n, f) != (size_t)n) { perror("fread"); return 2; }
			fclose(f);
			t.end = t.data.data() + n;
			const uint8_t *p = F.find_start_code(t.data.data(), t.end, 0);
			t.nal = p < t.end ? p + 3 : t.end;
			t.first_nal = t.nal; t.loops_left = loops - 1;
			t.dev_index = (int)((S.size() - 1) % devices.size());
			F.set_device(devices[(size_t)t.dev_index]); // the decoder is bound to the GPU selected when it is allocated
			t.dec = F.alloc(0, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
			if (!t.dec) { fprintf(stderr, "e264_multi: edge264_alloc failed (no MI355X / back end?)\n"); return 2; }
			if (!out_dir.empty()) {
				std::string o = out_dir + "/s" + std::to_string(S.size() - 1) + ".yuv";
				t.out = fopen(o.c_str(), "wb");
				if (!t.out) { perror(o.c_str()); return 2; }
			}
		}
	std::vector<void *> dev_obj(devices.size(), nullptr);
	if (!parse_only)
		for (Stream &s : S) {
			dev_obj[(size_t)s.dev_index] = F.device_of(s.dec);
			if (!dev_obj[(size_t)s.dev_index]) { fprintf(stderr, "e264_multi: no device\n"); return 2; }
		}
	FILE *dump = dump_path.empty() ? nullptr : fopen(dump_path.c_str(), "wb");

	auto drain = [&](Stream &s) {
		Edge264Frame fr;
		long n = 0;
		while (F.get_frame(s.dec, &fr, 0) == 0) {
			s.frames++; n++;
			if (s.out) write_frame(s.out, fr);
		}
		return n;
	};
	std::atomic<long> stuck_flushes{0}, stuck_given_up{0};
	long rounds = 0, packets = 0, total_frames = 0;
	// where the parser threads' time goes (summed over threads, seconds): inside edge264_decode_NAL, fetching frames, asleep with nothing to do
	std::atomic<long long> ns_decode{0}, ns_drain{0}, ns_idle{0}, ns_submit{0};
	auto now_ns = [] { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	std::mutex mu;                       // guards every Stream::q / finished and the counters below
	std::condition_variable cv_room;     // a packet left a queue (workers wait for room / for their packets to be on the device)
	std::condition_variable cv_ready;    // a packet entered a queue, or a decoder finished (the submitter waits)
	long queued = 0;
	// the output frames of a decoder may only be fetched when all its parsed pictures are on the device
	auto wait_submitted = [&](Stream &s) { std::unique_lock<std::mutex> lk(mu); cv_room.wait(lk, [&] { return s.q.empty(); }); };
	// 1. advance a decoder until its next picture is complete (or its stream ends).  Never waits for the device side: a decoder that
	//    cannot go on right now -- it is `ahead` pictures ahead, or it must hand out frames (ENOBUFS, edge264.h) while some of its
	//    pictures are still only queued -- reports BLOCKED and its thread turns to its next decoder (round 3; before, the thread
	//    slept until the next batch had taken the decoder's packet, with its other decoders idle)
	const bool must_submit_before_fetch = !parse_only && !no_download;
	enum Adv { GOT, ENDED, BLOCKED };
	auto advance = [&](Stream &s) -> Adv {
		{
			std::lock_guard<std::mutex> lk(mu);
			if ((int)s.q.size() >= ahead) return BLOCKED;
		}
		while (!s.done) {
			const uint8_t *nxt = s.nal < s.end ? F.find_start_code(s.nal, s.end, 0) : s.end;
			const long long t_dec = now_ns();
			int res = F.decode_NAL(s.dec, s.nal, nxt, nullptr, nullptr);
			ns_decode += now_ns() - t_dec;
			if (res == ENOBUFS) { // the same NAL again once frames have been fetched.  With read-back (edge264_get_frame downloads the picture)
				// they may only be fetched when all parsed pictures are on the device; without it (--no-download, --parse-only) fetching a
				// frame touches nothing on the device: the decoder goes on at once (round 4: before, every decoder stalled here until a
				// batch had taken its last packet, and the batches in turn waited for packets -- 4.6 k pictures/s where the same cores parse 9.3 k)
				if (must_submit_before_fetch) {
					std::lock_guard<std::mutex> lk(mu);
					if (!s.q.empty()) return BLOCKED;
				}
				const long long t_dr = now_ns();
				const long n_out = drain(s);
				ns_drain += now_ns() - t_dr;
				if (n_out == 0) {
					// ENOBUFS and nothing to hand out: a picture that will never complete (one of its slices failed and never came again) holds the
					// frame buffers, and the same NAL would return ENOBUFS for ever (until round 5 this loop did exactly that).  edge264_flush is
					// the API's way out (edge264.h:66: drop the delayed frames, clear the decoder state); decoding resumes at the next IDR picture.
					// A decoder that is still stuck right after a flush is given up.
					if (s.just_flushed) { s.done = true; stuck_given_up++; fprintf(stderr, "e264_multi: a decoder is still stuck right after edge264_flush: the rest of its stream is dropped\n"); break; }
					// (edge264_flush clears the pictures in progress: the packets this decoder has already queued must have left for the device first)
					if (!must_submit_before_fetch) {
						std::lock_guard<std::mutex> lk(mu);
						if (!s.q.empty()) return BLOCKED;
					}
					F.flush(s.dec);
					s.just_flushed = true;
					stuck_flushes++;
					drain(s);
					continue;
				}
				continue;
			}
			s.just_flushed = false;
			void *pkt = nullptr; size_t bytes = 0;
			bool got = F.take_packet(s.dec, &pkt, &bytes) == 0;
			if (res == ENODATA || s.nal >= s.end) s.done = true;
			else {
				s.nal = nxt + 3 < s.end ? nxt + 3 : s.end;
				if (s.nal >= s.end && s.loops_left > 0) { s.loops_left--; s.nal = s.first_nal; } // play it again (starts with SPS/PPS/IDR)
			}
			if (got) {
				std::lock_guard<std::mutex> lk(mu); // (room was checked on entry; only this thread adds to the queue)
				s.q.push_back({pkt, bytes});
				queued++;
				cv_ready.notify_all();
				return GOT;
			}
		}
		return ENDED;
	};
	if (n_threads < 1) n_threads = 1;
	if ((size_t)n_threads > S.size()) n_threads = (int)S.size();
	int workers_left = n_threads;
	// Any thread advances any decoder (round 4; before: thread k owned decoders k, k + T, ... -- with two input files and an even T half
	// the threads held only the CABAC B-picture stream, which parses at 460 pictures/s against 1150 for the CAVLC one, and the run
	// waited for them).  A thread walks the decoders from where it left off and takes the first one nobody holds that can go on.
	std::atomic<size_t> unfinished_count{S.size()};
	auto worker = [&](int k) {
		size_t at = (size_t)k * S.size() / (size_t)n_threads;
		while (unfinished_count.load() > 0) {
			bool progressed = false;
			for (size_t n = 0; n < S.size(); n++) {
				Stream &s = S[(at + n) % S.size()];
				if (s.finished || s.held.exchange(true)) continue;
				if (!s.finished) {
					const Adv a = advance(s);
					if (a == ENDED) { // its last packets still have to reach the device before its frames are fetched: wait_submitted below
						std::lock_guard<std::mutex> lk(mu);
						s.finished = true; unfinished_count--; cv_ready.notify_all();
					}
					// --stay: the thread comes back to THIS decoder first (its state is in the core's caches) until it is `ahead` pictures ahead;
					// default: on to the next decoder after every picture
					if (a == GOT) { progressed = true; at = (at + n + (stay ? 0 : 1)) % S.size(); s.held = false; break; }
				}
				s.held = false;
			}
			if (!progressed && unfinished_count.load() > 0) { // every decoder waits for a batch to take its packets (or is in another thread's hands)
				const long long t_id = now_ns();
				std::unique_lock<std::mutex> lk(mu);
				cv_room.wait_for(lk, std::chrono::milliseconds(1));
				ns_idle += now_ns() - t_id;
			}
		}
		for (size_t i = (size_t)k; i < S.size(); i += (size_t)n_threads) { wait_submitted(S[i]); drain(S[i]); }
		std::lock_guard<std::mutex> lk(mu);
		workers_left--;
		cv_ready.notify_all();
	};
	auto t0 = std::chrono::steady_clock::now();
	// (time, packets taken by device 0's submitter, ns inside decode_NAL) after every round: the STEADY rate is what comes after the first loop's worth of
	// pictures -- the first loop allocates every decoder's device frames, page-locked mirrors and packet buffers inside the clock (1.5 s for 128 decoders)
	struct Trail { double t; long packets; long long ns_decode; };
	std::vector<Trail> trail;
	std::vector<std::thread> pool;
	for (int k = 0; k < n_threads; k++) pool.emplace_back(worker, k);
	if (pin) {
		cpu_set_t allowed;
		if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
			std::vector<int> cpus;
			for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
			for (int k = 0; k < n_threads && !cpus.empty(); k++) {
				cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[(size_t)k % cpus.size()], &one);
				pthread_setaffinity_np(pool[(size_t)k].native_handle(), sizeof(one), &one);
			}
		}
	}
	// 2. the submitter: one batch = the oldest queued picture of every decoder that has one.  It waits until most of the
	//    decoders that are still parsing have a picture ready (or 2 ms have passed): a batch costs the device about one
	//    picture's dependency chain whatever its size, so small batches would only queue up device time.
	std::mutex dump_mu;
	auto submitter = [&](int di) {
	std::vector<void *> streams;
	std::vector<const void *> hpk;
	std::vector<size_t> hsz;
	std::vector<Stream *> owner;
	long my_packets = 0, my_rounds = 0;
	// packets submitted in place stay untouched until their batch has left the GPU: a ring of events (slots 8..15 of the device), the
	// packets of each round parked behind its event
	enum { NPEND = 4 };
	std::vector<void *> parked[NPEND];
	bool parked_busy[NPEND] = {};
	int pend_next = 0;
	// (--devices may name a GPU twice: its submitters share the device object and its 16 event slots; the first two get 4 slots
	// each, any further one submits pageable packets)
	int share = 0;
	for (int d = 0; d < di; d++) share += dev_obj[(size_t)d] == dev_obj[(size_t)di];
	const bool pinned = ::g_pinned && share < 2;
	if (share >= 2 && !parse_only) { fprintf(stderr, "e264_multi: a GPU may be named at most twice in --devices\n"); _exit(2); }
	const int ev0 = 8 + share * NPEND;
	auto release_round = [&](int k, bool block) {
		if (!parked_busy[k]) return true;
		while (H.event_query(dev_obj[(size_t)di], ev0 + k) != 0) {
			if (!block) return false;
			std::this_thread::sleep_for(std::chrono::microseconds(100));
		}
		for (void *p : parked[k]) F.free_packet(p);
		parked[k].clear();
		parked_busy[k] = false;
		return true;
	};
	for (;;) {
		streams.clear(); hpk.clear(); hsz.clear(); owner.clear();
		if (!parse_only) for (int k = 0; k < NPEND; k++) release_round(k, false);
		// Pacing by the DEVICE, not by a timer: at most two batches in flight (one in the kernels, one crossing PCIe behind it).  A batch
		// costs the GPU one picture's dependency chain whatever its size (~3 ms for 1 or 256 pictures), so while the batch before
		// runs, the parsers' packets pile up and the next batch takes ALL of them: batch size = what the cores parse in one batch
		// time, no tuning.  (Round 3 waited 2 ms or for 3/4 of the decoders: with 128 decoders on 16 cores neither ever came true
		// in time, batches of 11-19 pictures left every 2.4 ms and the decoders queued behind them.)
		if (!parse_only) release_round((pend_next + NPEND - 2) % NPEND, true);
		{
			std::unique_lock<std::mutex> lk(mu);
			auto ready = [&] { size_t n = 0; for (Stream &s : S) n += s.dev_index == di && !s.q.empty(); return n; };
			cv_ready.wait(lk, [&] { return ready() > 0 || workers_left == 0; });
			if (ready() == 0 && workers_left == 0) break;
			{ // a short pause so that a round is more than the first packet that turned up (an idle device, or none: --parse-only)
				auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(parse_only ? 500 : 300);
				cv_ready.wait_until(lk, deadline, [&] { return workers_left == 0; });
			}
			for (Stream &s : S)
				if (s.dev_index == di && !s.q.empty()) { owner.push_back(&s); hpk.push_back(s.q.front().data); hsz.push_back(s.q.front().bytes); }
		}
		for (size_t i = 0; i < owner.size(); i++) {
			if (dump) { std::lock_guard<std::mutex> dl(dump_mu); // capture file: the packets of all decoders interleaved, each tagged with its decoder (E264FrameHdr.stream_id, byte 76)
				uint32_t sid = (uint32_t)(owner[i] - &S[0]);
				memcpy((uint8_t *)hpk[i] + 76, &sid, 4);
				fwrite(hpk[i], 1, hsz[i], dump);
			}
			if (!parse_only) streams.push_back(F.stream(owner[i]->dec));
		}
		my_packets += (long)owner.size();
		my_rounds++;
		if (di == 0) trail.push_back({std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), my_packets, ns_decode.load()}); // (round: ~400 per second)
		const long long t_sub = now_ns();
		if (!parse_only) {
			release_round(pend_next, true);
			if (pinned) {
				if (H.submit_batch_pinned(dev_obj[(size_t)di], streams.data(), hpk.data(), hsz.data(), (int)streams.size(), 3, 1 /* E264_SUBMIT_TRUSTED: the front end ran e264hip_packet_check on them */)) { fprintf(stderr, "submit_batch_pinned: %s\n", H.last_error()); _exit(1); }
				for (const void *p : hpk) parked[pend_next].push_back(const_cast<void *>(p)); // untouched until the batch has left the GPU
			} else if (H.submit_batch_host(dev_obj[(size_t)di], streams.data(), hpk.data(), hsz.data(), (int)streams.size(), 3)) { fprintf(stderr, "submit_batch_host: %s\n", H.last_error()); _exit(1); }
			if (H.event_record(dev_obj[(size_t)di], ev0 + pend_next)) { fprintf(stderr, "event_record: %s\n", H.last_error()); _exit(1); }
			parked_busy[pend_next] = true;
			pend_next = (pend_next + 1) % NPEND;
		}
		ns_submit += now_ns() - t_sub;
		{ // the packets are on their way (copied to staging memory, or parked until their batch has retired): let their decoders go on
			std::lock_guard<std::mutex> lk(mu);
			for (Stream *s : owner) { if (!pinned) F.free_packet(s->q.front().data); s->q.pop_front(); }
			cv_room.notify_all();
		}
	}
	if (!parse_only) H.device_sync(dev_obj[(size_t)di]);
	for (int k = 0; k < NPEND; k++) release_round(k, true);
	std::lock_guard<std::mutex> lk(mu);
	packets += my_packets; rounds += my_rounds;
	};
	std::vector<std::thread> subs;
	for (size_t di = 1; di < devices.size(); di++) subs.emplace_back(submitter, (int)di);
	submitter(0);
	for (std::thread &t : subs) t.join();
	for (std::thread &t : pool) t.join();
	double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	for (Stream &s : S) {
		drain(s);
		total_frames += s.frames;
		if (s.out) fclose(s.out);
		F.free_dec(&s.dec);
	}
	if (dump) fclose(dump);
	// steady state: everything after the first loop's worth of packets (one device only; with fewer than two loops there is none)
	double st_fps = 0, st_ms = 0, st_after = 0;
	if (devices.size() == 1 && loops >= 2 && !trail.empty()) {
		const long warm = trail.back().packets / loops; // (--loops K: every stream is played K times)
		for (const Trail &w : trail)
			if (w.packets >= warm && warm > 0 && trail.back().packets > w.packets && trail.back().t > w.t) {
				st_after = w.t;
				st_fps = (double)(trail.back().packets - w.packets) / (trail.back().t - w.t);
				st_ms = (double)(ns_decode.load() - w.ns_decode) * 1e-6 / (double)(trail.back().packets - w.packets);
				break;
			}
	}
	printf("{\"streams\": %zu, \"devices\": %zu, \"threads\": %d, \"frames\": %ld, \"packets\": %ld, \"rounds\": %ld, \"avg_batch\": %.2f, \"seconds\": %.4f, \"frames_per_s\": %.1f, "
		"\"thread_seconds\": {\"decode_NAL\": %.2f, \"get_frame\": %.2f, \"idle\": %.2f, \"submit_calls\": %.2f}, \"decode_ms_per_picture\": %.3f, "
		"\"steady\": {\"after_seconds\": %.3f, \"frames_per_s\": %.1f, \"decode_ms_per_picture\": %.3f}, \"stuck_decoders_flushed\": %ld, \"stuck_decoders_given_up\": %ld}\n",
		S.size(), devices.size(), n_threads, total_frames, packets, rounds, rounds ? (double)packets / rounds : 0.0, sec, sec > 0 ? total_frames / sec : 0.0,
		ns_decode.load() * 1e-9, ns_drain.load() * 1e-9, ns_idle.load() * 1e-9, ns_submit.load() * 1e-9, packets ? ns_decode.load() * 1e-6 / (double)packets : 0.0,
		st_after, st_fps, st_ms, stuck_flushes.load(), stuck_given_up.load());
	return stuck_given_up.load() > 0 ? 3 : 0; // a stream that was cut short is not a clean run
}
