/*
 * fakehts.c -- a stand-in for libhts.so in the LD_PRELOAD test (the image has no
 * htslib/samtools).  What matters is the MECHANISM the reference relies on
 * (readme.md:9-14, bgzf_compress.c:39): libhts.so exports bgzf_compress() and its
 * own BGZF writer calls it through the PLT, so a preloaded library that exports
 * the same symbol takes the call over.  This file has the same shape: an exported
 * default bgzf_compress() and a writer in the same shared object that calls it
 * from worker threads, one 0xff00-byte block per call, members written in order.
 *
 * The default implementation here is deliberately not a compressor: it emits one
 * stored-block BGZF member, so the test can tell which implementation ran.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int bgzf_compress(void *_dst, size_t *dlen, const void *src, size_t slen, int level)
{
	/* 18-byte header, one stored block, CRC32 left zero on purpose, ISIZE */
	unsigned char *dst = _dst;
	(void)level;
	if (*dlen < slen + 18 + 5 + 8 || slen > 65535)
		return -1;
	static const unsigned char hdr[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
	memcpy(dst, hdr, 16);
	size_t total = 18 + 5 + slen + 8;
	dst[16] = (unsigned char)((total - 1) & 0xff);
	dst[17] = (unsigned char)((total - 1) >> 8);
	dst[18] = 1;
	dst[19] = slen & 0xff; dst[20] = slen >> 8; dst[21] = ~slen & 0xff; dst[22] = (~slen >> 8) & 0xff;
	memcpy(dst + 23, src, slen);
	memset(dst + 23 + slen, 0, 4);
	uint32_t n = (uint32_t)slen;
	memcpy(dst + 27 + slen, &n, 4);
	*dlen = total;
	return 0;
}

struct task {
	const unsigned char *src;
	size_t slen;
	unsigned char out[0x10000];
	size_t olen;
	int ret;
};

static struct task *g_tasks;
static int g_ntasks, g_next;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static void *worker(void *arg)
{
	(void)arg;
	for (;;) {
		pthread_mutex_lock(&g_mu);
		int i = g_next < g_ntasks ? g_next++ : -1;
		pthread_mutex_unlock(&g_mu);
		if (i < 0)
			return NULL;
		struct task *t = &g_tasks[i];
		t->olen = sizeof(t->out);                       /* htslib passes BGZF_MAX_BLOCK_SIZE = 0x10000 */
		t->ret = bgzf_compress(t->out, &t->olen, t->src, t->slen, -1);   /* through the PLT: interposable */
	}
}

/* the writer: what bgzf_write / the thread pool of hts do around bgzf_compress */
int fakehts_write_bgzf(FILE *out, const unsigned char *data, size_t n, int nthreads)
{
	g_ntasks = (int)((n + 0xff00 - 1) / 0xff00);
	g_next = 0;
	g_tasks = calloc((size_t)g_ntasks + 1, sizeof(struct task));
	for (int i = 0; i < g_ntasks; i++) {
		g_tasks[i].src = data + (size_t)i * 0xff00;
		g_tasks[i].slen = (size_t)i * 0xff00 + 0xff00 <= n ? 0xff00 : n - (size_t)i * 0xff00;
	}
	pthread_t th[64];
	if (nthreads > 64)
		nthreads = 64;
	for (int k = 0; k < nthreads; k++)
		pthread_create(&th[k], NULL, worker, NULL);
	for (int k = 0; k < nthreads; k++)
		pthread_join(th[k], NULL);
	int bad = 0;
	for (int i = 0; i < g_ntasks; i++) {
		if (g_tasks[i].ret)
			bad = 1;
		else
			fwrite(g_tasks[i].out, 1, g_tasks[i].olen, out);
	}
	/* the EOF marker comes from the same entry point, as in bgzf_flush/bgzf_close (slen = 0) */
	unsigned char eof[64];
	size_t el = sizeof(eof);
	int r = bgzf_compress(eof, &el, NULL, 0, -1);
	if (r == 0)
		fwrite(eof, 1, el, out);
	free(g_tasks);
	return bad;
}
