/*
 * hook_stress.c -- TEST ONLY: T threads x N calls of bgzf_compress() (bgzf_hook.c compiled with a sanitizer, linked
 * against stub_hipdeflate.c), every member checked against the block that went in; then the zlibutil_hip mirror
 * (hd_zlibutil_buffer_*) driven from threads the way applet/7bgzf.c:211 drives zlibutil_buffer_code; round 4: the codec
 * engines of the same batcher (hd_codec_batch) from the same T threads.
 *   hook_stress [threads=64] [calls=10000]
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hipdeflate.h"
#include "zlibutil_hip.h"

static int g_calls = 10000;
static int g_bad;

static void *worker(void *arg)
{
	const unsigned id = (unsigned)(uintptr_t)arg;
	unsigned char *src = (unsigned char *)malloc(0xff00), *dst = (unsigned char *)malloc(0x10000);
	unsigned seed = id * 2654435761u + 12345;
	for (int k = 0; k < g_calls; k++) {
		seed = seed * 1664525u + 1013904223u;
		const size_t n = (k % 97 == 0) ? 0 : 1 + (seed >> 8) % 0xff00;      /* now and then the EOF request */
		for (size_t i = 0; i < n; i += 61)
			src[i] = (unsigned char)(seed >> (i % 24));
		if (n) {
			src[n - 1] = (unsigned char)k;
			src[0] = (unsigned char)id;
		}
		size_t dlen = (k % 53 == 7) ? 20 : 0x10000;                          /* now and then too little room */
		const int r = bgzf_compress(dst, &dlen, src, n, -1);
		if (dlen == 20 && (k % 53 == 7)) {
			if (r != -1)
				__atomic_add_fetch(&g_bad, 1, __ATOMIC_RELAXED);
			continue;
		}
		if (r != 0 || (n == 0 && dlen != 28) ||
		    (n && (dlen != 18 + 5 + n + 8 || memcmp(dst + 23, src, n) || dst[23] != (unsigned char)id)))
			__atomic_add_fetch(&g_bad, 1, __ATOMIC_RELAXED);
	}
	free(src);
	free(dst);
	return NULL;
}

/* the per-block codecs' engines (bgzf_hook.c hd_codec_batch: one per level and raw frame), driven the same way; the stub's
 * contexts answer with the same stored BGZF-framed members whatever the frame */
int hd_codec_batch(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t slen, int level, int flush);
static void *codec_worker(void *arg)
{
	const unsigned id = (unsigned)(uintptr_t)arg;
	unsigned char *src = (unsigned char *)malloc(0xff00), *dst = (unsigned char *)malloc(0x18000);
	unsigned seed = id * 2246822519u + 777;
	for (int k = 0; k < g_calls / 4 + 1; k++) {
		seed = seed * 1664525u + 1013904223u;
		const size_t n = 1 + (seed >> 8) % 0xff00;
		for (size_t i = 0; i < n; i += 61)
			src[i] = (unsigned char)(seed >> (i % 24));
		src[n - 1] = (unsigned char)k;
		src[0] = (unsigned char)id;
		size_t dlen = 0x18000;
		const int r = hd_codec_batch(dst, &dlen, src, n, 1 + (int)(id % 3), (int)(id & 1));
		if (r != 0 || dlen != 18 + 5 + n + 8 || memcmp(dst + 23, src, n) || dst[23] != (unsigned char)id)
			__atomic_add_fetch(&g_bad, 1, __ATOMIC_RELAXED);
	}
	if (hd_codec_batch(dst, &(size_t){ 0x18000 }, src, 0, 1, 0) != -2 || hd_codec_batch(dst, &(size_t){ 0x18000 }, src, 0x10000, 1, 0) != -2)
		__atomic_add_fetch(&g_bad, 1, __ATOMIC_RELAXED);                   /* not for a batch: empty, or longer than a slot */
	free(src);
	free(dst);
	return NULL;
}

static void *zlibutil_worker(void *arg)
{
	const unsigned id = (unsigned)(uintptr_t)arg;
	for (int k = 0; k < 200; k++) {
		hd_zlibutil_buffer *zb = hd_zlibutil_buffer_allocate(6000, 4000 + id);
		if (!zb) {
			__atomic_add_fetch(&g_bad, 1, __ATOMIC_RELAXED);
			continue;
		}
		memset(zb->source, (int)(id + k), zb->sourceLen);
		zb->func = (void *)hip_deflate;
		zb->encode = 1;
		zb->level = 1;
		zb->rfc1952 = k & 1;
		zb->rfc1950 = !(k & 1) && (k & 2);
		hd_zlibutil_buffer_code(zb);
		if (zb->ret || zb->destLen < zb->sourceLen)
			__atomic_add_fetch(&g_bad, 1, __ATOMIC_RELAXED);
		hd_zlibutil_buffer_free(zb);
	}
	return NULL;
}

int main(int argc, char **argv)
{
	const int T = argc > 1 ? atoi(argv[1]) : 64;
	if (argc > 2)
		g_calls = atoi(argv[2]);
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)(T > 8 ? T : 8));
	for (int i = 0; i < T; i++)
		pthread_create(&th[i], NULL, worker, (void *)(uintptr_t)i);
	for (int i = 0; i < T; i++)
		pthread_join(th[i], NULL);
	for (int i = 0; i < T; i++)
		pthread_create(&th[i], NULL, codec_worker, (void *)(uintptr_t)i);
	for (int i = 0; i < T; i++)
		pthread_join(th[i], NULL);
	for (int i = 0; i < 8; i++)
		pthread_create(&th[i], NULL, zlibutil_worker, (void *)(uintptr_t)i);
	for (int i = 0; i < 8; i++)
		pthread_join(th[i], NULL);
	free(th);
	printf("hook_stress: %d threads x %d calls, %d bad\n", T, g_calls, g_bad);
	return g_bad ? 1 : 0;
}
