/*
 * hook_bench.c -- throughput of the LD_PRELOAD hook as htslib drives it: T threads, each calling
 * bgzf_compress() (bgzf_compress.c:39) on one 0xff00-byte block at a time and waiting for the member.
 *
 *   hook_bench <input file> [threads=8] [seconds=2] [block=65280] [codec:level]
 *
 * With a fifth argument the threads call that zlibutil codec (lib/zlibutil.h:47, e.g. libdeflate_deflate:1 or
 * hip_deflate:1, looked up with dlsym) instead of the hook: the per-block function alone, a fresh 1.5 x block
 * destination per call as zlibutil_buffer_allocate gives it -- bench.py's cpu_baseline runs the reference's
 * libdeflate_deflate this way, on real pthreads rather than through Python.  The name libdeflate_reused:<level> stands
 * for libdeflate_deflate_compress with ONE compressor per thread, allocated before the clock starts (SURVEY.md 8(d)(i):
 * the reference without lib/zlibutil.c:186-188's alloc/free per call).
 *
 * Links against whatever provides bgzf_compress: libhipdeflate.so (BGZF_METHOD=hip1) or the reference's
 * own hook built from bgzf_compress.c (oracle/_ref/libref.so, BGZF_METHOD=libdeflate1) -- the same binary
 * source measures both sides.  Prints one JSON line.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int bgzf_compress(void *dst, size_t *dlen, const void *src, size_t slen, int level);
#ifdef HOOK_BENCH_LAT
#include "hipdeflate.h"
#endif

typedef int (*codec_fn)(unsigned char *, size_t *, const unsigned char *, size_t, int);
static codec_fn g_codec;
static int g_codec_level = 1;
/* libdeflate_reused: the library's own entry points, looked up in whatever the binary is linked against */
static void *(*g_ld_alloc)(int);
static size_t (*g_ld_compress)(void *, const void *, size_t, void *, size_t);
static void (*g_ld_free)(void *);
static __thread void *t_comp;
static int reused_codec(unsigned char *dst, size_t *dlen, const unsigned char *src, size_t slen, int level)
{
	if (!t_comp && !(t_comp = g_ld_alloc(level)))
		return 2;
	const size_t n = g_ld_compress(t_comp, src, slen, dst, *dlen);
	*dlen = n;
	return n == 0;
}
static unsigned char *g_data;
static size_t g_size, g_block = 0xff00;
static volatile int g_stop;
static double g_secs = 2.0;

struct worker {
	pthread_t th;
	int id, nthreads;
	uint64_t in_bytes, out_bytes, calls;
	int err;
};

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static void *run(void *arg)
{
	struct worker *w = (struct worker *)arg;
	unsigned char *dst = (unsigned char *)malloc(g_block + g_block / 2 + 0x10000);
	const size_t nblk = g_size / g_block;
	size_t k = (size_t)w->id * 7919u % nblk;
	if (g_codec == reused_codec && !(t_comp = g_ld_alloc(g_codec_level)))
		w->err = 2;
	while (!g_stop && !w->err) {
		size_t dlen = g_codec ? g_block + g_block / 2 : 0x10000;
		int r = g_codec ? g_codec(dst, &dlen, g_data + k * g_block, g_block, g_codec_level)
				: bgzf_compress(dst, &dlen, g_data + k * g_block, g_block, -1);
		if (r) {
			w->err = r;
			break;
		}
		w->in_bytes += g_block;
		w->out_bytes += dlen;
		w->calls++;
		k = (k + (size_t)w->nthreads) % nblk;
	}
	if (t_comp) {
		g_ld_free(t_comp);
		t_comp = NULL;
	}
	free(dst);
	return NULL;
}

#ifdef HOOK_BENCH_LAT
/* HOOK_PAR=k (with threads = 0): k threads, each with its own latency context, run hipdeflate_lat_run on HOOK_N blocks side
 * by side -- what concurrent batches cost one another on the device and in the runtime, without the hook */
struct par {
	pthread_t th;
	int level, n;
	double us;
};
static void *par_run(void *arg)
{
	struct par *p = (struct par *)arg;
	hipdeflate_lat *c = hipdeflate_lat_open(p->level, HD_FRAME_BGZF | HD_FRAME_LATENCY, 256, 0xff00);
	uint32_t lens[256];
	if (!c)
		return NULL;
	for (int i = 0; i < p->n; i++) {
		memcpy(hipdeflate_lat_input(c, (uint32_t)i), g_data + (size_t)i * g_block, g_block);
		lens[i] = (uint32_t)g_block;
	}
	hipdeflate_lat_run(c, lens, (uint32_t)p->n);
	const int reps = 2000;
	const double t0 = now();
	for (int r = 0; r < reps; r++)
		if (hipdeflate_lat_run(c, lens, (uint32_t)p->n))
			break;
	p->us = (now() - t0) * 1e6 / reps;
	hipdeflate_lat_close(c);
	return NULL;
}
#endif

int main(int argc, char **argv)
{
	if (argc < 2) {
		fprintf(stderr, "usage: %s <input file> [threads] [seconds] [block]\n", argv[0]);
		return 2;
	}
	int T = argc > 2 ? atoi(argv[2]) : 8;
	if (argc > 3)
		g_secs = atof(argv[3]);
	if (argc > 4)
		g_block = (size_t)atol(argv[4]);
	const char *codec_name = "";
	if (argc > 5) {
		char name[128];
		snprintf(name, sizeof(name), "%s", argv[5]);
		char *c = strchr(name, ':');
		if (c) {
			*c = 0;
			g_codec_level = atoi(c + 1);
		}
		if (!strcmp(name, "libdeflate_reused")) {
			g_ld_alloc = (void *(*)(int))dlsym(RTLD_DEFAULT, "libdeflate_alloc_compressor");
			g_ld_compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(RTLD_DEFAULT, "libdeflate_deflate_compress");
			g_ld_free = (void (*)(void *))dlsym(RTLD_DEFAULT, "libdeflate_free_compressor");
			g_codec = g_ld_alloc && g_ld_compress && g_ld_free ? reused_codec : NULL;
		} else {
			g_codec = (codec_fn)dlsym(RTLD_DEFAULT, name);
		}
		if (!g_codec) {
			fprintf(stderr, "no such codec: %s\n", name);
			return 2;
		}
		codec_name = argv[5];
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f) {
		perror(argv[1]);
		return 2;
	}
	fseek(f, 0, SEEK_END);
	g_size = (size_t)ftell(f);
	fseek(f, 0, SEEK_SET);
	g_data = (unsigned char *)malloc(g_size);
	if (fread(g_data, 1, g_size, f) != g_size || g_size < g_block) {
		fprintf(stderr, "short input\n");
		return 2;
	}
	fclose(f);
#ifdef HOOK_BENCH_LAT
	if (T == 0) {
		/* the device side alone: hipdeflate_lat_run() on n blocks, one caller, no threads */
		const int level = getenv("HOOK_LEVEL") ? atoi(getenv("HOOK_LEVEL")) : 1;
		if (getenv("HOOK_PAR")) {
			const int k = atoi(getenv("HOOK_PAR")), n = getenv("HOOK_N") ? atoi(getenv("HOOK_N")) : 8;
			struct par ps[64];
			if (k < 1 || k > 64 || n < 1 || n > 256)
				return 2;
			for (int i = 0; i < k; i++) {
				ps[i].level = level;
				ps[i].n = n;
				ps[i].us = 0;
				pthread_create(&ps[i].th, NULL, par_run, &ps[i]);
			}
			double sum = 0;
			for (int i = 0; i < k; i++) {
				pthread_join(ps[i].th, NULL);
				sum += ps[i].us;
			}
			printf("{\"lat_run_parallel\": %d, \"blocks\": %d, \"level\": %d, \"us\": %.1f, \"GBps_in\": %.3f}\n", k, n, level,
			       sum / k, k * n * g_block / (sum / k) / 1e3);
			return 0;
		}
		hipdeflate_lat *c = hipdeflate_lat_open(level, HD_FRAME_BGZF | HD_FRAME_LATENCY, 256, 0xff00);
		if (!c)
			return 1;
		uint32_t lens[256];
		const int only = getenv("HOOK_N") ? atoi(getenv("HOOK_N")) : 0;  /* one batch size (for a kernel trace) */
		for (int n = 1; n <= 256; n *= 2) {
			if (only && n != only)
				continue;
			for (int i = 0; i < n; i++) {
				memcpy(hipdeflate_lat_input(c, (uint32_t)i), g_data + (size_t)i * g_block, g_block);
				lens[i] = (uint32_t)g_block;
			}
			hipdeflate_lat_run(c, lens, (uint32_t)n);
			const int reps = 200;
			/* HOOK_TOUCH=1: the input is written anew by the CPU before every run, as the hook's callers do (the time
			 * of the copy is not counted): what the device side costs when its input sits in the CPU's caches */
			const int touch = getenv("HOOK_TOUCH") != NULL;
			double sum = 0;
			for (int r = 0; r < reps; r++) {
				if (touch)
					for (int i = 0; i < n; i++)
						memcpy(hipdeflate_lat_input(c, (uint32_t)i), g_data + (size_t)((i + r) % 512) * g_block, g_block);
				const double t0 = now();
				if (hipdeflate_lat_run(c, lens, (uint32_t)n))
					return 1;
				sum += now() - t0;
			}
			const double us = sum * 1e6 / reps;
			printf("{\"lat_run_blocks\": %d, \"level\": %d, \"us\": %.1f, \"GBps_in\": %.3f}\n", n, level, us, n * g_block / us / 1e3);
		}
		hipdeflate_lat_close(c);
		return 0;
	}
#endif
	/* warm-up: first call initialises the device, pins memory ... */
	{
		unsigned char *dst = (unsigned char *)malloc(g_block + g_block / 2 + 0x10000);
		size_t dlen = g_codec ? g_block + g_block / 2 : 0x10000;
		int r = g_codec ? g_codec(dst, &dlen, g_data, g_block, g_codec_level) : bgzf_compress(dst, &dlen, g_data, g_block, -1);
		if (r) {
			fprintf(stderr, "bgzf_compress failed: %d\n", r);
			return 1;
		}
		if (t_comp) {
			g_ld_free(t_comp);
			t_comp = NULL;
		}
		free(dst);
	}
	struct worker *w = (struct worker *)calloc((size_t)T, sizeof(*w));
	const double t0 = now();
	for (int i = 0; i < T; i++) {
		w[i].id = i;
		w[i].nthreads = T;
		pthread_create(&w[i].th, NULL, run, &w[i]);
	}
	struct timespec d = { (time_t)g_secs, (long)((g_secs - (time_t)g_secs) * 1e9) };
	nanosleep(&d, NULL);
	g_stop = 1;
	uint64_t in = 0, out = 0, calls = 0;
	int err = 0;
	for (int i = 0; i < T; i++) {
		pthread_join(w[i].th, NULL);
		in += w[i].in_bytes;
		out += w[i].out_bytes;
		calls += w[i].calls;
		err |= w[i].err;
	}
	const double el = now() - t0;
	const char *m = getenv("BGZF_METHOD");
	if (g_codec)
		m = codec_name;
	printf("{\"threads\": %d, \"method\": \"%s\", \"block\": %zu, \"seconds\": %.3f, \"calls\": %llu, \"GBps_in\": %.4f, "
	       "\"ratio\": %.4f, \"us_per_call\": %.1f, \"error\": %d}\n",
	       T, m ? m : "", g_block, el, (unsigned long long)calls, in / el / 1e9, in ? (double)out / in : 0.0,
	       calls ? el * 1e6 * T / calls : 0.0, err);
	return err ? 1 : 0;
}
