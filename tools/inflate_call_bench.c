/* inflate_call_bench.c -- T threads calling hip_inflate() on one DEFLATE stream each, as `7bgzf -d -@T` does through
 * zlibutil_auto_inflate (applet/7bgzf.c:330-345: a thread per block): calls per second, latency per call, GB/s of output.
 *   inflate_call_bench STREAM OUT_LEN THREADS SECONDS      (STREAM: a file with one raw DEFLATE stream)
 * Built by 7bgzf_amd/csrc/Makefile as 7bgzf_amd/inflate_call_bench; tools/inflate_call_latency.py drives it. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "hipdeflate.h"

static unsigned char *g_z;
static size_t g_zn, g_out_len;
static volatile int g_stop;
static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
struct th { pthread_t t; long calls; int bad; double worst; };

static void *worker(void *arg)
{
	struct th *me = (struct th *)arg;
	unsigned char *dst = (unsigned char *)malloc(g_out_len + 64);
	unsigned char *src = (unsigned char *)malloc(g_zn + 8);
	memcpy(src, g_z, g_zn);
	memset(src + g_zn, 0xaa, 8);                                /* the member trailer the reference passes along, :328 */
	while (!g_stop) {
		size_t n = g_out_len;
		const double t0 = now_s();
		const int r = hip_inflate(dst, &n, src, g_zn + 8);
		const double dt = now_s() - t0;
		if (dt > me->worst)
			me->worst = dt;
		if (r || n != g_out_len)
			me->bad++;
		me->calls++;
	}
	free(dst);
	free(src);
	return NULL;
}

int main(int argc, char **argv)
{
	if (argc < 5) {
		fprintf(stderr, "usage: %s STREAM OUT_LEN THREADS SECONDS\n", argv[0]);
		return 2;
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	fseek(f, 0, SEEK_END);
	g_zn = (size_t)ftell(f);
	fseek(f, 0, SEEK_SET);
	g_z = (unsigned char *)malloc(g_zn);
	if (fread(g_z, 1, g_zn, f) != g_zn)
		return 2;
	fclose(f);
	g_out_len = (size_t)atol(argv[2]);
	const int T = atoi(argv[3]);
	const double secs = atof(argv[4]);
	if (hipdeflate_init(-1))
		return 1;
	{                                                           /* warm: contexts, pinned memory, code objects */
		unsigned char *dst = (unsigned char *)malloc(g_out_len + 64);
		for (int k = 0; k < 3; k++) {
			size_t n = g_out_len;
			if (hip_inflate(dst, &n, g_z, g_zn) || n != g_out_len) {
				fprintf(stderr, "hip_inflate failed on the stream\n");
				return 1;
			}
		}
		free(dst);
	}
	struct th *th = (struct th *)calloc((size_t)T, sizeof(*th));
	const double t0 = now_s();
	for (int i = 0; i < T; i++)
		pthread_create(&th[i].t, NULL, worker, &th[i]);
	struct timespec nap = { (time_t)secs, (long)((secs - (time_t)secs) * 1e9) };
	nanosleep(&nap, NULL);
	g_stop = 1;
	long calls = 0;
	int bad = 0;
	double worst = 0;
	for (int i = 0; i < T; i++) {
		pthread_join(th[i].t, NULL);
		calls += th[i].calls;
		bad += th[i].bad;
		if (th[i].worst > worst)
			worst = th[i].worst;
	}
	const double dt = now_s() - t0;
	printf("{\"threads\": %d, \"calls\": %ld, \"bad\": %d, \"us_per_call\": %.1f, \"worst_us\": %.0f, \"gbps_out\": %.4f, "
	       "\"devices\": %d}\n", T, calls, bad, dt / ((double)calls / T) * 1e6, worst * 1e6,
	       (double)calls * (double)g_out_len / dt / 1e9, hipdeflate_device_count());
	hipdeflate_shutdown();
	return bad != 0;
}
