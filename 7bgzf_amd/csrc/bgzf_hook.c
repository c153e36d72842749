/*
 * bgzf_hook.c -- the LD_PRELOAD hook: BGZF_METHOD=hip<level> LD_PRELOAD=./libhipdeflate.so samtools ...
 *
 * Same exported symbol, signature and return values as the reference's
 * bgzf_compress (bgzf_compress.c:39-198): htslib's own bgzf_compress is
 * shadowed through the PLT, so samtools/bcftools must link libhts.so
 * (readme.md:9-14).  Differences, all forced by the device:
 *
 *  - The only coder here is "hip" (BGZF_METHOD=hip, hip1 ... hip9; the trailing
 *    digits are the level, parsed as bgzf_compress.c:60-70 does; no digits = level 1,
 *    hip's default as each method has one at :102-112).  BGZF_METHOD unset or empty
 *    means that default too: preloading this library IS the choice of coder (the
 *    reference's default is its zlib at level 6, :54,:102).  A BGZF_METHOD that
 *    names anything else -- the reference's CPU coders, or an unknown name, which
 *    the reference silently runs as zlib -- returns -1, the reference's "coder
 *    missing" value (:136): this library holds no CPU codec, and coding with another
 *    method than the one NAMED would be the worse surprise.
 *  - htslib calls the hook once per 0xff00-byte block from each of its worker
 *    threads and waits for the member.  Calls that arrive together share one
 *    latency-mode batch (hipdeflate_lat_*, HD_FRAME_LATENCY: 16 wavefronts per block
 *    at level 1, members framed by the kernel: header, BSIZE, CRC32, ISIZE, i.e.
 *    bgzf_compress.c:191-197).  Every caller copies its own block into the batch's
 *    pinned memory and its own member out of it, outside any lock; the first caller
 *    of a batch (its leader) waits until the batch is as large as the previous one
 *    or HIPDEFLATE_BATCH_US microseconds (default 60) have passed, launches, and
 *    wakes the others.  HOOK_CTX batches can be in flight at once (one collecting,
 *    the others on the device).  A lone caller does not wait at all.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include "hipdeflate.h"
#include "hipdeflate_params.h"

#define HOOK_MAX_BATCH 256
#define HOOK_CTX 3
#define HOOK_BLOCK 0xff00u           /* what a latency-mode BGZF slot takes (16 x 4080); htslib's BGZF_BLOCK_SIZE */

struct hook_batch {
	hipdeflate_lat *lat;
	int state;                   /* 0 free, 1 collecting, 2 closed (waiting for copies / on the device), 3 done */
	int n, ready, taken;         /* blocks reserved, copied in, copied out */
	int rc;
	uint32_t len[HOOK_MAX_BATCH];
};

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;     /* any state change */
static struct hook_batch g_batch[HOOK_CTX];
static int g_open = -1;       /* the batch that is collecting, -1 = none */
static int g_last_batch = 1;  /* size of the previous batch: what a leader waits for */
static int g_method = -1;     /* -1 unparsed, 0 not ours, 1 hip */
static int g_level = 1;
static long g_window_us = 60;
static int g_batch_target = HOOK_MAX_BATCH;
static int g_failed;

static void parse_env(void)
{
	/* bgzf_compress.c:53-113: name = prefix, level = trailing decimal digits */
	const char *s = getenv("BGZF_METHOD");
	g_method = 1;                       /* unset / empty: this library's one coder */
	g_level = 1;
	if (s && *s) {
		size_t l = strlen(s), i = l;
		int level = -1, digit = 1;
		while (i > 0 && s[i - 1] >= '0' && s[i - 1] <= '9') {
			if (level < 0)
				level = 0;
			level += digit * (s[i - 1] - '0');
			digit *= 10;
			i--;
		}
		if (i == 3 && !strncasecmp(s, "hip", 3))
			g_level = level >= 0 ? level : 1;
		else
			g_method = 0;
	}
	const char *w = getenv("HIPDEFLATE_BATCH_US");
	if (w && *w)
		g_window_us = atol(w);
	const char *t = getenv("HIPDEFLATE_BATCH_BLOCKS");
	if (t && *t) {
		g_batch_target = atoi(t);
		if (g_batch_target < 1)
			g_batch_target = 1;
		if (g_batch_target > HOOK_MAX_BATCH)
			g_batch_target = HOOK_MAX_BATCH;
	}
}

/* a block the latency slots do not take (longer than 0xff00 bytes: not from htslib): one ordinary call */
static int code_alone(void *dst, size_t *dlen, const void *src, size_t slen)
{
	uint64_t off = 0;
	uint32_t len = (uint32_t)slen, olen = 0;
	int32_t st = 0;
	unsigned char *tmp = (unsigned char *)malloc(65536);
	if (!tmp)
		return -1;
	int rc = hipdeflate_batch_deflate((const uint8_t *)src, &off, &len, 1, g_level, HD_FRAME_BGZF, tmp, 65536, 65536, &olen,
					  NULL, &st);
	int ret = rc ? -1 : (st || olen > *dlen) ? 1 : 0;
	if (ret == 1)
		fprintf(stderr, "hip_deflate %d\n", st ? st : 1);
	if (!ret) {
		memcpy(dst, tmp, olen);
		*dlen = olen;
	}
	free(tmp);
	return ret;
}

int bgzf_compress(void *_dst, size_t *_dlen, const void *src, size_t slen, int level_unused)
{
	(void)level_unused;
	if (!slen) {
		/* bgzf_compress.c:40-51 */
		if (*_dlen < 28)
			return -1;
		*_dlen = 28;
		memcpy(_dst,
		       "\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff"
		       "\x06\0BC\x02\x00"
		       "\x1b\x00"
		       "\x03\x00"
		       "\x00\x00\x00\x00\x00\x00\x00\x00",
		       28);
		return 0;
	}
	pthread_mutex_lock(&g_mu);
	if (g_method < 0)
		parse_env();
	if (g_method != 1) {
		pthread_mutex_unlock(&g_mu);
		fprintf(stderr, "hipdeflate: BGZF_METHOD must be hip<level> (or unset); no other coder in this library\n");
		return -1;
	}
	if (*_dlen < 26) {                          /* bgzf_compress.c:116 */
		pthread_mutex_unlock(&g_mu);
		return -1;
	}
	if (slen > 0x10000) {                       /* a BGZF member cannot hold it */
		pthread_mutex_unlock(&g_mu);
		return 1;
	}
	if (slen > HOOK_BLOCK || g_failed) {
		const int failed = g_failed;
		pthread_mutex_unlock(&g_mu);
		return failed ? -1 : code_alone(_dst, _dlen, src, slen);
	}
	/* ---- join the collecting batch, or open one ---------------------------------------------- */
	struct hook_batch *b;
	for (;;) {
		if (g_open >= 0) {
			b = &g_batch[g_open];
			break;
		}
		int k;
		for (k = 0; k < HOOK_CTX && g_batch[k].state != 0; k++)
			;
		if (k < HOOK_CTX) {
			b = &g_batch[k];
			if (!b->lat) {
				b->lat = hipdeflate_lat_open(g_level, HD_FRAME_BGZF | HD_FRAME_LATENCY, HOOK_MAX_BATCH, HOOK_BLOCK);
				if (!b->lat) {
					g_failed = 1;
					pthread_mutex_unlock(&g_mu);
					return -1;                          /* coder missing */
				}
			}
			b->state = 1;
			b->n = b->ready = b->taken = 0;
			g_open = k;
			break;
		}
		pthread_cond_wait(&g_cv, &g_mu);        /* every context is busy: wait for one to drain */
	}
	const int idx = b->n++;
	const int leader = idx == 0;
	b->len[idx] = (uint32_t)slen;
	if (b->n >= g_batch_target || b->n >= HOOK_MAX_BATCH || (!leader && b->n >= g_last_batch)) {
		b->state = 2;                               /* full, or as large as the last one: close it */
		g_open = -1;
		pthread_cond_broadcast(&g_cv);
	}
	pthread_mutex_unlock(&g_mu);

	memcpy(hipdeflate_lat_input(b->lat, (uint32_t)idx), src, slen);      /* own block, no lock held */

	pthread_mutex_lock(&g_mu);
	b->ready++;
	if (leader) {
		if (b->state == 1 && g_window_us > 0 && g_last_batch > 1) {
			struct timespec ts;
			clock_gettime(CLOCK_REALTIME, &ts);
			ts.tv_nsec += g_window_us * 1000L;
			ts.tv_sec += ts.tv_nsec / 1000000000L;
			ts.tv_nsec %= 1000000000L;
			while (b->state == 1)
				if (pthread_cond_timedwait(&g_cv, &g_mu, &ts) == ETIMEDOUT)
					break;
		}
		if (b->state == 1) {                        /* window over (or nobody to wait for) */
			b->state = 2;
			g_open = -1;
		}
		while (b->ready < b->n)                     /* the others are still copying in */
			pthread_cond_wait(&g_cv, &g_mu);
		const int n = b->n;
		g_last_batch = n;
		pthread_mutex_unlock(&g_mu);
		const int rc = hipdeflate_lat_run(b->lat, b->len, (uint32_t)n);
		pthread_mutex_lock(&g_mu);
		b->rc = rc;
		b->state = 3;
		pthread_cond_broadcast(&g_cv);
	} else {
		if (b->ready == b->n)
			pthread_cond_broadcast(&g_cv);          /* the leader may be waiting for this copy */
		while (b->state != 3)
			pthread_cond_wait(&g_cv, &g_mu);
	}
	const int rc = b->rc;
	pthread_mutex_unlock(&g_mu);

	int ret;
	uint32_t olen = 0;
	int32_t st = 0;
	const uint8_t *m = hipdeflate_lat_output(b->lat, (uint32_t)idx, &olen, NULL, &st);
	if (rc || !m) {
		ret = -1;                                   /* coder missing */
	} else if (st || olen > *_dlen) {
		fprintf(stderr, "hip_deflate %d\n", st ? st : 1);
		ret = 1;                                    /* codec error, bgzf_compress.c:163-169 */
	} else {
		memcpy(_dst, m, olen);                      /* own member, no lock held */
		*_dlen = olen;
		ret = 0;
	}
	pthread_mutex_lock(&g_mu);
	if (++b->taken == b->n) {
		b->state = 0;                               /* drained: the context can collect again */
		pthread_cond_broadcast(&g_cv);
	}
	pthread_mutex_unlock(&g_mu);
	return ret;
}
