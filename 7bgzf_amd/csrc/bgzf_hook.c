/*
 * bgzf_hook.c -- the LD_PRELOAD hook: BGZF_METHOD=hip<level> LD_PRELOAD=./libhipdeflate.so samtools ...
 *
 * Same exported symbol, signature and return values as the reference's
 * bgzf_compress (bgzf_compress.c:39-198): htslib's own bgzf_compress is
 * shadowed through the PLT, so samtools/bcftools must link libhts.so
 * (readme.md:9-14).  Differences, all forced by the device:
 *
 *  - the only method served is "hip" (BGZF_METHOD=hip, hip1 ... hip9; the
 *    trailing digits are the level, parsed as bgzf_compress.c:60-70 does).
 *    Anything else returns -1, the reference's "coder missing" value
 *    (bgzf_compress.c:136): this library holds no CPU codec to fall back to.
 *  - htslib calls the hook once per 0xff00-byte block from each of its worker
 *    threads.  One block per launch cannot feed a GPU, so concurrent calls are
 *    micro-batched: the first caller becomes the leader, waits up to
 *    HIPDEFLATE_BATCH_US microseconds (default 200) for the other workers'
 *    blocks, and compresses them all with ONE hipdeflate_batch_deflate call in
 *    HD_FRAME_BGZF mode (the kernel writes header, BSIZE, CRC32 and ISIZE, i.e.
 *    bgzf_compress.c:191-197).  A lone caller (batch of 1 last time) does not
 *    wait at all.  The call stays synchronous, as the reference's is.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include "hipdeflate.h"

#define HOOK_MAX_BATCH 256
#define HOOK_SLOT 65536

struct hook_req {
	const void *src;
	size_t slen;
	void *dst;
	size_t cap;
	size_t out;
	int ret;
	int done;
};

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_full = PTHREAD_COND_INITIALIZER;   /* queue reached the batch size */
static pthread_cond_t g_done = PTHREAD_COND_INITIALIZER;   /* a batch finished */
static struct hook_req *g_queue[HOOK_MAX_BATCH];
static int g_qn;
static int g_leader;          /* a leader is collecting */
static int g_last_batch = 1;  /* size of the previous batch: 1 => do not wait */
static int g_method = -1;     /* -1 unparsed, 0 not ours, 1 hip */
static int g_level = 1;
static long g_window_us = 200;
static int g_batch_target = 64;

static void parse_env(void)
{
	/* bgzf_compress.c:53-113: name = prefix, level = trailing decimal digits */
	const char *s = getenv("BGZF_METHOD");
	g_method = 0;
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
		if (i == 3 && !strncasecmp(s, "hip", 3)) {
			g_method = 1;
			g_level = level >= 0 ? level : 1;
		}
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

/* compress `n` queued requests with one launch; called WITHOUT g_mu held */
static void run_batch(struct hook_req **reqs, int n)
{
	static __thread unsigned char *in_buf, *out_buf;
	static __thread size_t in_cap, out_cap;
	uint64_t off[HOOK_MAX_BATCH];
	uint32_t len[HOOK_MAX_BATCH], olen[HOOK_MAX_BATCH];
	int32_t st[HOOK_MAX_BATCH];
	size_t total = 0;
	for (int i = 0; i < n; i++) {
		off[i] = total;
		len[i] = (uint32_t)reqs[i]->slen;
		total += (reqs[i]->slen + 15) & ~(size_t)15;
	}
	if (total > in_cap) {
		free(in_buf);
		in_buf = (unsigned char *)malloc(in_cap = total + 65536);
	}
	if ((size_t)n * HOOK_SLOT > out_cap) {
		free(out_buf);
		out_buf = (unsigned char *)malloc(out_cap = (size_t)n * HOOK_SLOT);
	}
	int rc = HD_E_NOMEM;
	if (in_buf && out_buf) {
		for (int i = 0; i < n; i++)
			memcpy(in_buf + off[i], reqs[i]->src, reqs[i]->slen);
		rc = hipdeflate_batch_deflate(in_buf, off, len, (uint32_t)n, g_level, HD_FRAME_BGZF, out_buf, HOOK_SLOT,
					      HOOK_SLOT, olen, NULL, st);
	}
	for (int i = 0; i < n; i++) {
		struct hook_req *r = reqs[i];
		if (rc) {
			r->ret = -1;                    /* coder missing */
		} else if (st[i] || olen[i] > r->cap) {
			fprintf(stderr, "hip_deflate %d\n", st[i] ? st[i] : 1);
			r->ret = 1;                     /* codec error, bgzf_compress.c:163-169 */
		} else {
			memcpy(r->dst, out_buf + (size_t)i * HOOK_SLOT, olen[i]);
			r->out = olen[i];
			r->ret = 0;
		}
	}
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
		fprintf(stderr, "hipdeflate: BGZF_METHOD must be hip<level>; no other coder in this library\n");
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
	struct hook_req me = { src, slen, _dst, *_dlen, 0, 0, 0 };
	while (g_qn >= HOOK_MAX_BATCH)              /* queue full: wait for a batch to drain */
		pthread_cond_wait(&g_done, &g_mu);
	g_queue[g_qn++] = &me;
	if (g_qn >= g_batch_target)
		pthread_cond_signal(&g_full);
	if (!g_leader) {
		g_leader = 1;
		if (g_window_us > 0 && (g_last_batch > 1 || g_qn > 1)) {
			struct timespec ts;
			clock_gettime(CLOCK_REALTIME, &ts);
			ts.tv_nsec += g_window_us * 1000L;
			ts.tv_sec += ts.tv_nsec / 1000000000L;
			ts.tv_nsec %= 1000000000L;
			while (g_qn < g_batch_target)
				if (pthread_cond_timedwait(&g_full, &g_mu, &ts) == ETIMEDOUT)
					break;
		}
		struct hook_req *batch[HOOK_MAX_BATCH];
		int n = g_qn;
		memcpy(batch, g_queue, sizeof(batch[0]) * (size_t)n);
		g_qn = 0;
		g_leader = 0;
		g_last_batch = n;
		pthread_mutex_unlock(&g_mu);
		run_batch(batch, n);
		pthread_mutex_lock(&g_mu);
		for (int i = 0; i < n; i++)
			batch[i]->done = 1;
		pthread_cond_broadcast(&g_done);
	}
	while (!me.done)
		pthread_cond_wait(&g_done, &g_mu);
	pthread_mutex_unlock(&g_mu);
	if (me.ret == 0)
		*_dlen = me.out;
	return me.ret;
}
