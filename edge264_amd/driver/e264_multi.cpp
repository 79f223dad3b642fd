// e264_multi -- batched multi-stream front end (the N-stream counterpart of the
// reference's `edge264_test -b`, test.c:427-546).
//
// N decoder instances (edge264.h API, served by the reference's parsers + our emitters in sink mode 2: frames
// live in HBM, finished command packets are queued) are advanced round-robin, one frame each per round; the
// packets of a round go to the GPU as ONE batch (e264hip_submit_batch: 4 kernel launches for all streams), then
// every decoder's output frames are fetched.  Host parsing (the reference front end with n_threads = 0) is spread
// over --threads T host threads, each owning a fixed subset of the decoders; the batch submission is done by the
// main thread between two barriers.
//
//   e264_multi --front <libedge264_hipfront.so> --hip <libedge264_hip.so> [--device N] [--repeat R]
//              [--threads T] [--out DIR] [--dump-packets FILE] [--parse-only] [--no-download] a.264 b.264 ...
//
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
#include <signal.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
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
	const uint8_t *(*find_start_code)(const uint8_t *, const uint8_t *, int);
	void (*set_sink)(int);
	void (*set_device)(int);
	void (*set_download)(int);
	int (*take_packet)(void *, void **, size_t *);
	void (*free_packet)(void *);
	void *(*stream)(void *);
	void *(*device)(void);
};
struct Hip {
	int (*submit_batch_host)(void *, void **, const void **, const size_t *, int, int);
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
	void *pkt = nullptr; size_t pkt_bytes = 0;
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

int main(int argc, char **argv)
{
	signal(SIGSEGV, on_crash);
	signal(SIGBUS, on_crash);
	std::string front_path, hip_path, out_dir, dump_path;
	int device = 0, repeat = 1, n_threads = 1, loops = 1; // --loops K: every stream is played K times back to back (steady state)
	bool no_download = false; // --no-download: output frames stay in HBM (edge264_get_frame does not copy them back)
	bool parse_only = false; // --parse-only: sink 1, no GPU: packets are produced and dropped (front-end speed / debugging)
	std::vector<std::string> files;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto next = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
		if (a == "--front") front_path = next();
		else if (a == "--hip") hip_path = next();
		else if (a == "--device") device = atoi(next().c_str());
		else if (a == "--repeat") repeat = atoi(next().c_str());
		else if (a == "--parse-only") parse_only = true;
		else if (a == "--no-download") no_download = true;
		else if (a == "--threads") n_threads = atoi(next().c_str());
		else if (a == "--loops") loops = atoi(next().c_str());
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
	bind(fl, "edge264_free", F.free_dec); bind(fl, "edge264_find_start_code", F.find_start_code);
	bind(fl, "e264front_set_sink", F.set_sink); bind(fl, "e264front_set_device", F.set_device); bind(fl, "e264front_set_download", F.set_download);
	bind(fl, "e264front_take_packet", F.take_packet); bind(fl, "e264front_free_packet", F.free_packet);
	bind(fl, "e264front_stream", F.stream); bind(fl, "e264front_device", F.device);
	if (!parse_only) {
	bind(hl, "e264hip_submit_batch_host", H.submit_batch_host); bind(hl, "e264hip_device_sync", H.device_sync);
	bind(hl, "e264hip_last_error", H.last_error);
	}

	F.set_device(device);
	F.set_sink(parse_only ? 1 : 2);
	F.set_download(no_download ? 0 : 1);
	std::vector<Stream> S;
	for (int r = 0; r < repeat; r++)
		for (const std::string &path : files) {
			Stream s;
			FILE *f = fopen(path.c_str(), "rb");
			if (!f) { perror(path.c_str()); return 2; }
			fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
			s.data.resize((size_t)n + 64);
			if (fread(s.data.data(), 1, (size_t)n, f) != (size_t)n) { perror("fread"); return 2; }
			fclose(f);
			S.push_back(std::move(s));
			Stream &t = S.back();
			t.end = t.data.data() + n;
			const uint8_t *p = F.find_start_code(t.data.data(), t.end, 0);
			t.nal = p < t.end ? p + 3 : t.end;
			t.first_nal = t.nal; t.loops_left = loops - 1;
			t.dec = F.alloc(0, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
			if (!t.dec) { fprintf(stderr, "e264_multi: edge264_alloc failed (no MI355X / back end?)\n"); return 2; }
			if (!out_dir.empty()) {
				std::string o = out_dir + "/s" + std::to_string(S.size() - 1) + ".yuv";
				t.out = fopen(o.c_str(), "wb");
				if (!t.out) { perror(o.c_str()); return 2; }
			}
		}
	void *dev = parse_only ? nullptr : F.device();
	if (!dev && !parse_only) { fprintf(stderr, "e264_multi: no device\n"); return 2; }
	FILE *dump = dump_path.empty() ? nullptr : fopen(dump_path.c_str(), "wb");

	auto drain = [&](Stream &s) {
		Edge264Frame fr;
		while (F.get_frame(s.dec, &fr, 0) == 0) {
			s.frames++;
			if (s.out) write_frame(s.out, fr);
		}
	};
	long rounds = 0, packets = 0, total_frames = 0;
	std::vector<void *> streams;
	std::vector<const void *> hpk;
	std::vector<size_t> hsz;
	// 1. advance a decoder until its next frame is complete (or its stream ends)
	auto advance = [&](Stream &s) {
		while (!s.done && !s.pkt) {
			const uint8_t *nxt = s.nal < s.end ? F.find_start_code(s.nal, s.end, 0) : s.end;
			int res = F.decode_NAL(s.dec, s.nal, nxt, nullptr, nullptr);
			if (res == ENOBUFS) { drain(s); continue; } // every earlier packet of this stream is already on the device
			if (F.take_packet(s.dec, &s.pkt, &s.pkt_bytes) != 0) s.pkt = nullptr;
			if (res == ENODATA || s.nal >= s.end) { s.done = true; break; }
			s.nal = nxt + 3 < s.end ? nxt + 3 : s.end;
			if (s.nal >= s.end && s.loops_left > 0) { s.loops_left--; s.nal = s.first_nal; } // play it again (starts with SPS/PPS/IDR)
		}
	};
	// worker threads: thread k owns decoders k, k+T, k+2T, ...; phases are separated by a counting barrier
	if (n_threads < 1) n_threads = 1;
	if ((size_t)n_threads > S.size()) n_threads = (int)S.size();
	std::mutex mu; std::condition_variable cv;
	int phase = 0, arrived = 0; bool quit = false;
	auto barrier = [&](std::unique_lock<std::mutex> &lk) { // all n_threads participants (main is participant 0)
		int my = phase;
		if (++arrived == n_threads) { arrived = 0; phase++; cv.notify_all(); }
		else cv.wait(lk, [&] { return phase != my; });
	};
	std::vector<std::thread> pool;
	for (int k = 1; k < n_threads; k++)
		pool.emplace_back([&, k] {
			for (;;) {
				{ std::unique_lock<std::mutex> lk(mu); barrier(lk); if (quit) return; } // start of a parse phase
				for (size_t i = (size_t)k; i < S.size(); i += (size_t)n_threads) { drain(S[i]); advance(S[i]); }
				{ std::unique_lock<std::mutex> lk(mu); barrier(lk); }                    // end of the parse phase
			}
		});
	auto t0 = std::chrono::steady_clock::now();
	for (;;) {
		bool any = false;
		{ std::unique_lock<std::mutex> lk(mu); barrier(lk); }
		for (size_t i = 0; i < S.size(); i += (size_t)n_threads) { drain(S[i]); advance(S[i]); }
		{ std::unique_lock<std::mutex> lk(mu); barrier(lk); }
		for (Stream &s : S) any |= s.pkt != nullptr;
		// 2. one batch for the whole round: staged + copied + launched asynchronously; the next round's parsing overlaps
		//    with it, and edge264_get_frame (download) or the final sync is where the host meets the device again
		streams.clear(); hpk.clear(); hsz.clear();
		for (Stream &s : S)
			if (s.pkt) {
				if (dump) { // capture file: the packets of all decoders interleaved, each tagged with its decoder (E264FrameHdr.stream_id, byte 76)
					uint32_t sid = (uint32_t)(&s - &S[0]);
					memcpy((uint8_t *)s.pkt + 76, &sid, 4);
					fwrite(s.pkt, 1, s.pkt_bytes, dump);
				}
				packets++;
				if (!parse_only) { streams.push_back(F.stream(s.dec)); hpk.push_back(s.pkt); hsz.push_back(s.pkt_bytes); }
			}
		if (!streams.empty()) {
			if (H.submit_batch_host(dev, streams.data(), hpk.data(), hsz.data(), (int)streams.size(), 3)) { fprintf(stderr, "submit_batch_host: %s\n", H.last_error()); return 1; }
			rounds++;
		}
		for (Stream &s : S)
			if (s.pkt) { F.free_packet(s.pkt); s.pkt = nullptr; }
		if (parse_only && any) rounds++;
		// 3. output happens at the start of the next parse phase (each thread drains its own decoders)
		if (!any) break;
	}
	if (!parse_only) H.device_sync(dev);
	{ std::unique_lock<std::mutex> lk(mu); quit = true; barrier(lk); }
	for (std::thread &t : pool) t.join();
	double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	for (Stream &s : S) {
		drain(s);
		total_frames += s.frames;
		if (s.out) fclose(s.out);
		F.free_dec(&s.dec);
	}
	if (dump) fclose(dump);
	printf("{\"streams\": %zu, \"threads\": %d, \"frames\": %ld, \"packets\": %ld, \"rounds\": %ld, \"avg_batch\": %.2f, \"seconds\": %.4f, \"frames_per_s\": %.1f}\n",
		S.size(), n_threads, total_frames, packets, rounds, rounds ? (double)packets / rounds : 0.0, sec, sec > 0 ? total_frames / sec : 0.0);
	return 0;
}
