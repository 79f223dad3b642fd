/* tools/hostprof/hostprof.c -- MEASUREMENT TOOL (not product): what one host core spends per 1080p picture in
 *   (a) the unmodified reference decoder              oracle/_ref/libedge264_ref.so       (parse + reconstruct + deblock)
 *   (b) the reference's parser with NULL leaves       tools/hostprof/libedge264_nullfront.so (the parse floor: the sample
 *       kernels replaced by stubs that only clear ctx->c, nothing emitted)
 *   (c) the product front end, capture sink           edge264_amd/libedge264_hipfront.so   (parse + emit + packet assembly)
 * each decoding the given Annex-B files round-robin on ONE pinned core for a fixed time, like src/edge264_test.c:482-542
 * times the reference (allocation included).  Libraries are bound with dlopen through the 7 functions of edge264.h:64-70.
 *   hostprof LIB seconds core a.264 [b.264 ...]
 * With hipfront the packets are taken and freed after every NAL (e264front_take_packet), i.e. the cost of assembling
 * them is inside the measurement; their sizes are reported. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void *(*alloc_t)(int, void *, void *, int, void *, void *, void *);
typedef void (*free_t)(void **);
typedef const uint8_t *(*find_t)(const uint8_t *, const uint8_t *, int);
typedef int (*decode_t)(void *, const uint8_t *, const uint8_t *, void *, void *);
typedef int (*get_t)(void *, void *, int);
typedef int (*take_t)(void *, void **, size_t *);
typedef void (*freepkt_t)(void *);
typedef void (*setsink_t)(int);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static double cpu(void) { struct timespec t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: hostprof LIB seconds core a.264 ...\n"); return 2; }
	const double seconds = atof(argv[2]);
	const int core = atoi(argv[3]);
	if (core >= 0) { cpu_set_t s; CPU_ZERO(&s); CPU_SET(core, &s); sched_setaffinity(0, sizeof(s), &s); }
	void *L = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
	if (!L) { fprintf(stderr, "%s\n", dlerror()); return 2; }
	alloc_t A = (alloc_t)dlsym(L, "edge264_alloc"); free_t F = (free_t)dlsym(L, "edge264_free");
	find_t S = (find_t)dlsym(L, "edge264_find_start_code"); decode_t D = (decode_t)dlsym(L, "edge264_decode_NAL");
	get_t G = (get_t)dlsym(L, "edge264_get_frame");
	take_t T = (take_t)dlsym(L, "e264front_take_packet"); freepkt_t FP = (freepkt_t)dlsym(L, "e264front_free_packet");
	setsink_t SS = (setsink_t)dlsym(L, "e264front_set_sink");
	if (!A || !F || !S || !D || !G) { fprintf(stderr, "not an edge264.h library\n"); return 2; }
	if (SS) SS(1); /* capture sink: no device */
	int nf = argc - 4;
	uint8_t **buf = calloc((size_t)nf, sizeof(*buf)); size_t *len = calloc((size_t)nf, sizeof(*len));
	for (int i = 0; i < nf; i++) {
		FILE *f = fopen(argv[4 + i], "rb");
		if (!f) { perror(argv[4 + i]); return 2; }
		fseek(f, 0, SEEK_END); len[i] = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
		buf[i] = calloc(1, len[i] + 64);
		if (fread(buf[i], 1, len[i], f) != len[i]) return 2;
		fclose(f);
	}
	uint8_t frame[512];
	long frames = 0, packets = 0; double pkt_bytes = 0;
	const double t0 = now(), c0 = cpu();
	for (int k = 0; now() - t0 < seconds; k++) {
		const uint8_t *base = buf[k % nf], *end = base + len[k % nf];
		void *dec = A(0, NULL, NULL, 0, NULL, NULL, NULL);
		if (!dec) { fprintf(stderr, "edge264_alloc failed\n"); return 1; }
		const uint8_t *nal = S(base, end, 0);
		nal = nal < end ? nal + 3 : end;
		for (;;) {
			const uint8_t *nxt = nal < end ? S(nal, end, 0) : end;
			int res = D(dec, nal, nxt, NULL, NULL), got = 0;
			if (T) { void *p; size_t n; while (T(dec, &p, &n) == 0) { packets++; pkt_bytes += (double)n; FP(p); } }
			while (G(dec, frame, 0) == 0) got++;
			frames += got;
			if (res == ENOBUFS) { if (!got) break; continue; }
			if (res == ENODATA || nal >= end) break;
			nal = nxt + 3 < end ? nxt + 3 : end;
		}
		while (G(dec, frame, 0) == 0) frames++;
		F(&dec);
	}
	const double w = now() - t0, c = cpu() - c0;
	printf("{\"lib\": \"%s\", \"frames\": %ld, \"wall_s\": %.3f, \"cpu_s\": %.3f, \"frames_per_s\": %.1f, \"core_ms_per_picture\": %.3f, \"packets\": %ld, \"mean_packet_bytes\": %.0f}\n",
		argv[1], frames, w, c, frames / w, frames ? c * 1e3 / frames : 0.0, packets, packets ? pkt_bytes / packets : 0.0);
	for (int i = 0; i < nf; i++) free(buf[i]);
	free(buf); free(len); /* (tools/sanitize runs this driver under the leak checker) */
	return 0;
}
