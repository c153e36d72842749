/*
 * hd_bgzf_host.c -- hd7bgzf: the per-block loop of applet/7bgzf.c (_compress
 * :133-293, _decompress :295-365) and applet/7migz.c re-shaped into a BATCH loop
 * over libhipdeflate.so.  Same stdin -> stdout filter behaviour and the same
 * stderr lines ("compression level = N (hip)", "N done.", "ellapsed time") so
 * scripts written for `7bgzf` keep working:
 *
 *     hd7bgzf -G1 < in > out.bgz        (-G<level> or -l<level>; level 0..9)
 *     hd7bgzf -d  < in.bgz > out
 *     hd7bgzf -M -b1024 -G6 < in > out.migz    (MiGz framing, block = b KiB)
 *     hd7bgzf -G1 --index out.bgz.gzi < in > out.bgz     (+ bgzip's .gzi block index)
 *     hd7bgzf -G1 -@8 -i in -o out.bgz                    (file to file: N threads pread / pwrite)
 *     hd7bgzf -G1 -g8 -i in -o out.bgz                    (-g N: the batches are dealt round robin to N devices -- the entries
 *                                                          of HIPDEFLATE_DEVICES if set, else devices 0..N-1; the reference's
 *                                                          analogue is -@ N worker threads, applet/7bgzf.c:155-217)
 *
 * What changed, and why: the reference reads one <=64 KiB block, compresses it on
 * a fresh pthread and writes it (applet/7bgzf.c:159-277); a GPU needs thousands
 * of blocks per launch, so this host drives a hipdeflate_pipe: batches of HD_BATCH
 * blocks are read straight into pinned memory, three of them are in flight (H2D
 * copy, kernels in HD_FRAME_BGZF mode -- the kernel writes header, BSIZE, CRC32,
 * ISIZE --, device-side gather, D2H copy), and every finished batch is one
 * contiguous run of members that leaves with a single write().
 * Blocks are 0xff00 bytes (the reference's multi-thread size, :146-147); the
 * single-thread 0x10000 + shrink-by-1024 retry (:256-262) is not needed because
 * the kernel falls back to stored blocks, which always fit.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include "hipdeflate.h"

#define HD_BATCH 4096          /* decode: members per launch */
#define HD_PIPE_BATCH 512      /* encode: blocks per pipe batch (32 MiB of pinned input each; pinning memory costs ~0.3 ms per MiB) */

static const unsigned char eof_member[28] = { 0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0x00, 0xff, 0x06, 0x00, 'B',
					      'C',  0x02, 0x00, 0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0 };

static uint32_t rd16(const unsigned char *p) { return p[0] | (p[1] << 8); }
static uint32_t rd32(const unsigned char *p) { return rd16(p) | (rd16(p + 2) << 16); }

static size_t g_block = 0xff00;
static int g_frame = HD_FRAME_BGZF;

/* --index FILE: bgzip's .gzi -- u64 count, then (compressed offset, uncompressed offset) u64 pairs, little-endian,
 * one per block start except the first (htslib bgzf_index_dump: "noffs - 1" records when writing; no record for
 * the EOF member).  The compressed offsets are the device's size prefix scan of every batch + the bytes written
 * before it: nothing walks the members on the host.  BAM virtual offsets are HIPDEFLATE_VOFFSET(coffset, uoffset). */
static const char *g_index_path;
static uint64_t *g_index;          /* pairs */
static size_t g_index_n, g_index_cap;

static int index_add(uint64_t coff, uint64_t uoff)
{
	if (g_index_n == g_index_cap) {
		g_index_cap = g_index_cap ? g_index_cap * 2 : 4096;
		uint64_t *q = (uint64_t *)realloc(g_index, g_index_cap * 16);
		if (!q)
			return 1;
		g_index = q;
	}
	g_index[2 * g_index_n] = coff;
	g_index[2 * g_index_n + 1] = uoff;
	g_index_n++;
	return 0;
}

static int index_write(void)
{
	FILE *f = fopen(g_index_path, "wb");
	if (!f)
		return 1;
	const uint64_t cnt = g_index_n;        /* (this host runs on little-endian machines only: x86-64 next to an MI355X) */
	int bad = fwrite(&cnt, 8, 1, f) != 1 || (g_index_n && fwrite(g_index, 16, g_index_n, f) != g_index_n);
	bad |= fclose(f) != 0;
	return bad;
}

/* compress: the reader thread fills pinned batches straight from stdin and submits
 * them, the main thread takes the finished batches (one contiguous run of members
 * each) in order and writes them; three batches are in flight on the device side */
/* -g N: one pipe per entry of the device list; batch j goes to pipe j % N, the results are fetched in the same order,
 * so the one reader and the one in-order writer of the single-device host drive N cards and the output is byte for
 * byte that of -g 1 (blocks are independent) */
#define HD_MAX_G 32
static hipdeflate_pipe *g_pipes[HD_MAX_G];
static int g_npipe = 1;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
static int g_submitted, g_reader_done, g_reader_err;

static void *reader_main(void *arg)
{
	(void)arg;
	for (int j = 0;; j++) {
		hipdeflate_pipe *g_pipe = g_pipes[j % g_npipe];
		size_t cap = 0, got = 0;
		unsigned char *buf = hipdeflate_pipe_input(g_pipe, &cap);
		int err = buf == NULL;
		while (!err && got < cap) {
			ssize_t r = read(0, buf + got, cap - got);
			if (r < 0 && errno == EINTR)
				continue;
			if (r <= 0) {
				err = r < 0 ? 2 : 0;
				break;
			}
			got += (size_t)r;
		}
		if (!err && hipdeflate_pipe_submit(g_pipe, got))
			err = 1;
		pthread_mutex_lock(&mu);
		if (err)
			g_reader_err = err;
		else
			g_submitted++;
		if (err || got < cap)
			g_reader_done = 1;
		pthread_cond_broadcast(&cv);
		pthread_mutex_unlock(&mu);
		if (err || got < cap)
			return NULL;
	}
}

static int write_all(const unsigned char *p, size_t n)
{
	while (n) {
		ssize_t w = write(1, p, n);
		if (w < 0 && errno == EINTR)
			continue;
		if (w <= 0)
			return 1;
		p += w;
		n -= (size_t)w;
	}
	return 0;
}

static int open_pipes(int level, uint32_t per, int depth)
{
	const int ndev = hipdeflate_device_count();
	for (int k = 0; k < g_npipe; k++) {
		g_pipes[k] = hipdeflate_pipe_open_on(k % (ndev > 0 ? ndev : 1), level, g_frame, (uint32_t)g_block, per, depth);
		if (!g_pipes[k]) {
			fprintf(stderr, "hip_deflate: cannot open the device pipeline\n");
			return 1;
		}
	}
	return 0;
}

static int do_compress(int level)
{
	size_t per = g_frame == HD_FRAME_MIGZ ? (HD_PIPE_BATCH * (size_t)0xff00 + g_block - 1) / g_block : HD_PIPE_BATCH;
	if (getenv("HD7BGZF_BATCH") && atol(getenv("HD7BGZF_BATCH")) > 0)
		per = (size_t)atol(getenv("HD7BGZF_BATCH"));
	if (open_pipes(level, (uint32_t)per, g_npipe > 1 ? 3 : 4))
		return 1;
	pthread_t rd;
	pthread_create(&rd, NULL, reader_main, NULL);
	int total_blocks = 0, chk = 64, fetched = 0, ret = 0;
	uint64_t written = 0;
	for (;;) {
		pthread_mutex_lock(&mu);
		while (fetched == g_submitted && !g_reader_done)
			pthread_cond_wait(&cv, &mu);
		int more = fetched < g_submitted, err = g_reader_err;
		pthread_mutex_unlock(&mu);
		if (!more) {
			if (err) {
				fprintf(stderr, err == 2 ? "read error\n" : "hip_deflate: submit failed\n");
				ret = 1;
			}
			break;
		}
		const uint8_t *data;
		size_t nbytes;
		uint32_t nb;
		hipdeflate_pipe *g_pipe = g_pipes[fetched % g_npipe];
		int r = hipdeflate_pipe_result(g_pipe, &data, &nbytes, &nb);
		if (r) {
			fprintf(stderr, "hip_deflate %d\n", r);                      /* applet/7bgzf.c:228-254 */
			ret = 1;
			break;
		}
		fetched++;
		if (g_index_path) {
			const uint64_t *doff;
			if (hipdeflate_pipe_members(g_pipe, NULL, &doff, NULL)) {
				ret = 1;
				break;
			}
			for (uint32_t i = 0; i < nb; i++)
				if (total_blocks + (int)i > 0 && index_add(written + doff[i], (uint64_t)(total_blocks + (int)i) * g_block)) {
					ret = 1;
					break;
				}
		}
		if (write_all(data, nbytes)) {
			fprintf(stderr, "write error\n");
			ret = 1;
			break;
		}
		written += nbytes;
		total_blocks += (int)nb;
		while (total_blocks >= chk) {                     /* progress as applet/7bgzf.c:278-281 */
			fprintf(stderr, "%d\r", chk);
			chk += 64;
		}
	}
	if (ret)
		_exit(1);                                         /* the reader may sit in read(): do not join it */
	pthread_join(rd, NULL);
	for (int k = 0; k < g_npipe; k++)
		hipdeflate_pipe_close(g_pipes[k]);
	if (g_frame == HD_FRAME_BGZF && write_all(eof_member, 28))            /* applet/7bgzf.c:283-289 */
		return 1;
	if (g_index_path && index_write()) {
		fprintf(stderr, "cannot write %s\n", g_index_path);
		return 1;
	}
	fprintf(stderr, "%d done.\n", total_blocks);
	return 0;
}

/* ---- file to file: -i IN -o OUT [-@ N] ------------------------------------------------------------------------
 * The stdin/stdout filter above moves ~5 GB/s: one thread read()s, and pinning 32 MiB buffers is paid per run.  With
 * both ends seekable the host side scales: N worker threads pread() disjoint 8 MiB ranges of a batch straight into
 * the pipe's pinned input (page cache -> pinned memory at memcpy speed on every thread), batches are up to 8192
 * blocks so that the device is full, and the finished run of members is pwrite()n by the same workers in disjoint
 * ranges at the offset the batches before it ended (the device's size scan).  Role of the read / code / write loop
 * of applet/7bgzf.c:159-293 with its -@ threads, minus the thread per block. */
struct io_task {
	int fd, wr;
	unsigned char *buf;
	size_t len;
	off_t off;
};
/* two channels with their own workers, so that reading batch k + 1 and writing batch k - 1 overlap: 0 = pread, 1 = pwrite */
static struct io_chan {
	struct io_task task[256];
	int n, next, done, err, stop;
	pthread_mutex_t mu;
	pthread_cond_t cv, done_cv;
} g_io[2] = { { .mu = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER, .done_cv = PTHREAD_COND_INITIALIZER },
	      { .mu = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER, .done_cv = PTHREAD_COND_INITIALIZER } };

static void *io_worker(void *arg)
{
	struct io_chan *c = (struct io_chan *)arg;
	pthread_mutex_lock(&c->mu);
	for (;;) {
		while (!c->stop && c->next >= c->n)
			pthread_cond_wait(&c->cv, &c->mu);
		if (c->stop)
			break;
		const struct io_task t = c->task[c->next++];
		pthread_mutex_unlock(&c->mu);
		int err = 0;
		for (size_t done = 0; done < t.len && !err;) {
			const ssize_t r = t.wr ? pwrite(t.fd, t.buf + done, t.len - done, t.off + (off_t)done)
					       : pread(t.fd, t.buf + done, t.len - done, t.off + (off_t)done);
			if (r < 0 && errno == EINTR)
				continue;
			if (r <= 0)
				err = 1;
			else
				done += (size_t)r;
		}
		pthread_mutex_lock(&c->mu);
		c->err |= err;
		if (++c->done == c->n)
			pthread_cond_broadcast(&c->done_cv);
	}
	pthread_mutex_unlock(&c->mu);
	return NULL;
}

/* move [off, off + len) between the file and buf in parallel pieces (one caller per channel) */
static int io_parallel(int fd, int wr, unsigned char *buf, size_t len, off_t off)
{
	struct io_chan *c = &g_io[wr];
	size_t piece = (size_t)8 << 20;
	if (len / piece >= 256)
		piece = (len / 255 + 4095) & ~(size_t)4095;
	pthread_mutex_lock(&c->mu);
	c->n = 0;
	for (size_t o = 0; o < len; o += piece) {
		struct io_task *t = &c->task[c->n++];
		t->fd = fd;
		t->wr = wr;
		t->buf = buf + o;
		t->off = off + (off_t)o;
		t->len = len - o < piece ? len - o : piece;
	}
	c->next = c->done = c->err = 0;
	pthread_cond_broadcast(&c->cv);
	while (c->n && c->done < c->n)
		pthread_cond_wait(&c->done_cv, &c->mu);
	const int err = c->err;
	c->n = c->next = 0;
	pthread_mutex_unlock(&c->mu);
	return err;
}

static int g_fd_in = -1, g_fd_out = -1;
static off_t g_in_size;
static double g_t_ctor, g_t_main;       /* HD7BGZF_TIMING: when the process' constructors ran, when main() began */
static int g_timing;                 /* HD7BGZF_TIMING=1: where the wall time of the file-to-file path goes (stderr) */
static double g_t_read, g_t_write, g_t_result, g_t_open, g_t_input;

static double now_s(void)
{
	struct timeval tv;
	gettimeofday(&tv, NULL);
	return tv.tv_sec + tv.tv_usec * 1e-6;
}

static void *file_reader_main(void *arg)
{
	(void)arg;
	int j = 0;
	for (off_t pos = 0;; j++) {
		hipdeflate_pipe *g_pipe = g_pipes[j % g_npipe];
		size_t cap = 0;
		double t0 = now_s();
		unsigned char *buf = hipdeflate_pipe_input(g_pipe, &cap);
		g_t_input += now_s() - t0;
		int err = buf == NULL;
		const size_t got = (off_t)cap < g_in_size - pos ? cap : (size_t)(g_in_size - pos);
		t0 = now_s();
		if (!err && got && io_parallel(g_fd_in, 0, buf, got, pos))
			err = 2;
		g_t_read += now_s() - t0;
		pos += (off_t)got;
		if (!err && hipdeflate_pipe_submit(g_pipe, got))
			err = 1;
		const int last = err || pos >= g_in_size;
		pthread_mutex_lock(&mu);
		if (err)
			g_reader_err = err;
		else
			g_submitted++;
		if (last)
			g_reader_done = 1;
		pthread_cond_broadcast(&cv);
		pthread_mutex_unlock(&mu);
		if (last)
			return NULL;
	}
}

static void *fallocate_main(void *arg)
{
	(void)arg;
	const off_t want = g_in_size / 2 + (1 << 20);           /* more than half is rare for what people compress; the rest grows on write */
	for (off_t o = 0; o < want; o += (off_t)256 << 20)
		if (fallocate(g_fd_out, FALLOC_FL_KEEP_SIZE, o, want - o < ((off_t)256 << 20) ? want - o : (off_t)256 << 20))
			break;                                            /* not supported here: the writers allocate as they go */
	return NULL;
}

static int do_compress_files(int level, int nthreads)
{
	const size_t nblocks = (size_t)((g_in_size + (off_t)g_block - 1) / (off_t)g_block);
	/* Batches of 2048 blocks (128 MiB): this path is bound by the host's reads, writes and by pinning memory (~0.3 ms per
	 * MiB, paid when a slot is first used), not by the kernels, so a batch only needs to be large enough for the copies and
	 * launches to overlap -- 8192-block batches (rounds 2-3) pinned 3 GiB before the first byte moved.  Smaller for small
	 * files so that a few are in flight per device.  HD7BGZF_BATCH overrides. */
	/* (round 5: a sixteenth of the file, not a quarter -- a 256 MiB file went through four batches of 64 MiB, each in a slot
	 * of its own: 512 MiB pinned, every slot used once, 0.15 s of the job's 0.25; now each slot is used four times) */
	size_t per = nblocks / (size_t)(16 * g_npipe) + 1;
	if (per > 2048)
		per = 2048;
	if (getenv("HD7BGZF_BATCH") && atol(getenv("HD7BGZF_BATCH")) > 0)
		per = (size_t)atol(getenv("HD7BGZF_BATCH"));
	if (g_frame == HD_FRAME_MIGZ && per * g_block > ((size_t)512 << 20))
		per = ((size_t)512 << 20) / g_block;
	if (per < 64)
		per = 64;
	double t0 = now_s();
	if (open_pipes(level, (uint32_t)per, g_npipe > 1 ? 3 : 4))
		return 1;
	g_t_open = now_s() - t0;
	/* -@ N: N readers and N writers.  The output's pages are allocated ahead of the writers by a helper thread
	 * (fallocate of what the data is likely to need, in 256 MiB steps: a first write into a fresh page-cache or tmpfs
	 * page costs several times a memcpy -- measured 5 GB/s for 8 writers without it) */
	const int nwr = nthreads;
	pthread_t fa;
	pthread_create(&fa, NULL, fallocate_main, NULL);
	pthread_t *io = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)(nthreads + nwr));
	for (int i = 0; i < nthreads + nwr; i++)
		pthread_create(&io[i], NULL, io_worker, &g_io[i >= nthreads]);
	pthread_t rd;
	pthread_create(&rd, NULL, file_reader_main, NULL);
	int total_blocks = 0, fetched = 0, ret = 0;
	off_t written = 0;
	for (;;) {
		pthread_mutex_lock(&mu);
		while (fetched == g_submitted && !g_reader_done)
			pthread_cond_wait(&cv, &mu);
		const int more = fetched < g_submitted, err = g_reader_err;
		pthread_mutex_unlock(&mu);
		if (!more) {
			if (err) {
				fprintf(stderr, err == 2 ? "read error\n" : "hip_deflate: submit failed\n");
				ret = 1;
			}
			break;
		}
		const uint8_t *data;
		size_t nbytes;
		uint32_t nb;
		t0 = now_s();
		hipdeflate_pipe *g_pipe = g_pipes[fetched % g_npipe];
		const int r = hipdeflate_pipe_result(g_pipe, &data, &nbytes, &nb);
		g_t_result += now_s() - t0;
		if (r) {
			fprintf(stderr, "hip_deflate %d\n", r);
			ret = 1;
			break;
		}
		fetched++;
		if (g_index_path) {
			const uint64_t *doff;
			if (hipdeflate_pipe_members(g_pipe, NULL, &doff, NULL)) {
				ret = 1;
				break;
			}
			for (uint32_t i = 0; i < nb && !ret; i++)
				if (total_blocks + (int)i > 0 &&
				    index_add((uint64_t)written + doff[i], (uint64_t)(total_blocks + (int)i) * g_block))
					ret = 1;
			if (ret)
				break;
		}
		t0 = now_s();
		if (nbytes && io_parallel(g_fd_out, 1, (unsigned char *)(uintptr_t)data, nbytes, written)) {
			fprintf(stderr, "write error\n");
			ret = 1;
			break;
		}
		g_t_write += now_s() - t0;
		written += (off_t)nbytes;
		total_blocks += (int)nb;
	}
	if (ret)
		_exit(1);
	pthread_join(rd, NULL);
	for (int k = 0; k < 2; k++) {
		pthread_mutex_lock(&g_io[k].mu);
		g_io[k].stop = 1;
		pthread_cond_broadcast(&g_io[k].cv);
		pthread_mutex_unlock(&g_io[k].mu);
	}
	for (int i = 0; i < nthreads + nwr; i++)
		pthread_join(io[i], NULL);
	pthread_join(fa, NULL);
	struct stat so;
	if (!fstat(g_fd_out, &so) && S_ISREG(so.st_mode) && ftruncate(g_fd_out, written + (g_frame == HD_FRAME_BGZF ? 28 : 0)))
		return 1;
	free(io);
	for (int k = 0; k < g_npipe; k++)
		hipdeflate_pipe_close(g_pipes[k]);
	if (g_frame == HD_FRAME_BGZF && pwrite(g_fd_out, eof_member, 28, written) != 28)        /* applet/7bgzf.c:283-289 */
		return 1;
	if (g_index_path && index_write()) {
		fprintf(stderr, "cannot write %s\n", g_index_path);
		return 1;
	}
	if (g_timing)
		fprintf(stderr, "timing: batches of %zu blocks; open pipeline (pinning) %.3f s; reader: waits for a free batch %.3f, pread %.3f; "
			"writer: waits for results %.3f, pwrite %.3f\n", per, g_t_open, g_t_input, g_t_read, g_t_result, g_t_write);
	fprintf(stderr, "%d done.\n", total_blocks);
	return 0;
}


/* header walk of _read_gz_header (applet/7bgzf.c:81-131): FLG / XLEN / FNAME / FCOMMENT / FHCRC, then the
 * member length from the extra field -- BC (BGZF, u16 + 1), MZ (MiGz, payload u32 + header + 8), IG v1
 * (mgzip, u64 whole member), IG v2 (u32 whole member), jerodsanto's mgzip (u24 whole member, 0x7d tag).
 * 1 = ok: *hdr = header bytes, *total = whole member; always hdr + 8 <= total (room for CRC32 + ISIZE),
 * so the payload length and the ISIZE read stay inside the member.  0 = not such a member / cut off. */
static int member_len(const unsigned char *p, size_t avail, size_t *hdr, size_t *total)
{
	if (avail < 12 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0) || !(p[3] & 4))
		return 0;
	const uint32_t xlen = rd16(p + 10);
	const unsigned char *x = p + 12;
	if (avail < 12 + (size_t)xlen)
		return 0;
	size_t n = 12 + xlen;
	if (p[3] & 0x08) { while (n < avail && p[n++]) ; }
	if (p[3] & 0x10) { while (n < avail && p[n++]) ; }
	if (p[3] & 0x02) n += 2;
	if (n > avail)
		return 0;
	uint64_t t;
	if (xlen == 6 && !memcmp(x, "BC\x02\x00", 4))
		t = (uint64_t)rd16(x + 4) + 1;
	else if (xlen == 8 && !memcmp(x, "MZ\x04\x00", 4))
		t = (uint64_t)rd32(x + 4) + n + 8;
	else if (xlen == 20 && !memcmp(x, "IG\x10\x00", 4))
		t = (uint64_t)rd32(x + 4) | (uint64_t)rd32(x + 8) << 32;
	else if (xlen == 8 && !memcmp(x, "IG\x04\x00", 4))
		t = rd32(x + 4);
	else if (xlen == 4 && x[3] == 0x7d)
		t = rd32(x) & 0xffffffu;
	else
		return 0;
	if (t < n + 8 || t > 0xfffffff0u)
		return 0;
	*hdr = n;
	*total = (size_t)t;
	return 1;
}

/* decompress: the reader thread reads compressed bytes straight into pinned batches, walks
 * the member headers there (the serial part: applet/7bgzf.c:306-327), carries the cut-off
 * member at the end over to the next batch and submits the table; the main thread writes each
 * finished batch -- one contiguous run of output -- in order */
#define UNP_IN_CAP  ((size_t)16 << 20)
#define UNP_OUT_CAP ((size_t)96 << 20)
static size_t g_unp_in_cap = UNP_IN_CAP, g_unp_out_cap = UNP_OUT_CAP;     /* what the slots were opened with (do_decompress) */
static hipdeflate_unpipe *g_unpipes[HD_MAX_G];
static int g_members;

static void reader_fail(int code)
{
	pthread_mutex_lock(&mu);
	g_reader_err = code;
	g_reader_done = 1;
	pthread_cond_broadcast(&cv);
	pthread_mutex_unlock(&mu);
}

static void *unreader_main(void *arg)
{
	(void)arg;
	static uint64_t ioff[HD_BATCH];
	static uint32_t ilen[HD_BATCH], osz[HD_BATCH];
	unsigned char *carry = NULL;
	size_t carry_len = 0, carry_cap = 0;
	int eof = 0;
	for (int j = 0; !eof || carry_len; j++) {
		hipdeflate_unpipe *g_unpipe = g_unpipes[j % g_npipe];
		size_t cap = 0;
		unsigned char *buf = hipdeflate_unpipe_input(g_unpipe, &cap);
		if (!buf || carry_len > cap) {
			reader_fail(1);
			return NULL;
		}
		size_t have = carry_len;
		if (carry_len)
			memcpy(buf, carry, carry_len);
		carry_len = 0;
		while (!eof && have < cap) {
			ssize_t r = read(0, buf + have, cap - have);
			if (r < 0 && errno == EINTR)
				continue;
			if (r < 0) {
				reader_fail(2);
				return NULL;
			}
			if (r == 0)
				eof = 1;
			have += (size_t)r;
		}
		uint32_t nb = 0;
		size_t p = 0, osum = 0;
		while (nb < HD_BATCH && p < have) {
			size_t hdr, total;
			if (!member_len(buf + p, have - p, &hdr, &total)) {
				/* a header the end of the batch cut (FNAME / IG v1 headers pass 20 bytes): carry it
				 * over and look again with more bytes; p > 0 guarantees progress */
				if (!eof && p > 0 && have - p < 4096)
					break;
				reader_fail(3);
				return NULL;
			}
			if (p + total > have) {
				if (eof || total > cap) {
					reader_fail(3);
					return NULL;
				}
				break;
			}
			const uint32_t isize = rd32(buf + p + total - 4);
			if (osum + isize > g_unp_out_cap) {
				if (!nb) {
					reader_fail(3);
					return NULL;
				}
				break;
			}
			ioff[nb] = p + hdr;
			ilen[nb] = (uint32_t)(total - hdr);                 /* payload + 8-byte trailer, :328 */
			osz[nb] = isize;
			osum += isize;
			nb++;
			p += total;
		}
		if (p < have) {                                         /* the cut-off tail goes first in the next batch */
			carry_len = have - p;
			if (carry_len > carry_cap) {
				free(carry);
				carry = malloc(carry_cap = carry_len + (1u << 20));
			}
			memcpy(carry, buf + p, carry_len);
		}
		if (!nb && carry_len && eof) {
			reader_fail(3);
			return NULL;
		}
		if (hipdeflate_unpipe_submit(g_unpipe, ioff, ilen, osz, nb)) {
			reader_fail(1);
			return NULL;
		}
		pthread_mutex_lock(&mu);
		g_submitted++;
		g_members += (int)nb;
		pthread_cond_broadcast(&cv);
		pthread_mutex_unlock(&mu);
	}
	free(carry);
	pthread_mutex_lock(&mu);
	g_reader_done = 1;
	pthread_cond_broadcast(&cv);
	pthread_mutex_unlock(&mu);
	return NULL;
}

static int do_decompress(void)
{
	const int ndev = hipdeflate_device_count();
	/* a small file: smaller slots, each used several times, instead of 3 x 112 MiB pinned for one use each (pinning is ~0.3 ms
	 * per MiB, paid when a slot is first used) */
	g_unp_in_cap = UNP_IN_CAP;
	g_unp_out_cap = UNP_OUT_CAP;
	struct stat si;
	unsigned char h18[18];
	/* (BGZF only -- its members are at most 64 KiB; the other member kinds may need the large slot for ONE member) */
	if (!fstat(0, &si) && S_ISREG(si.st_mode) && si.st_size > 0 && (uint64_t)si.st_size < ((uint64_t)1 << 30) &&
	    pread(0, h18, 18, 0) == 18 && h18[0] == 0x1f && h18[1] == 0x8b && (h18[3] & 4) && h18[12] == 'B' && h18[13] == 'C') {
		size_t oc = ((size_t)si.st_size * 4 / (12 * (size_t)g_npipe) + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
		if (oc < ((size_t)8 << 20))
			oc = (size_t)8 << 20;
		if (oc < g_unp_out_cap) {
			g_unp_out_cap = oc;
			g_unp_in_cap = oc / 4 < ((size_t)4 << 20) ? (size_t)4 << 20 : oc / 4;
		}
	}
	for (int k = 0; k < g_npipe; k++) {
		g_unpipes[k] = hipdeflate_unpipe_open_on(k % (ndev > 0 ? ndev : 1), HD_BATCH, g_unp_in_cap, g_unp_out_cap, 3);
		if (!g_unpipes[k]) {
			fprintf(stderr, "inflate: cannot open the device pipeline\n");
			return 1;
		}
	}
	pthread_t rd;
	pthread_create(&rd, NULL, unreader_main, NULL);
	int fetched = 0, ret = 0;
	for (;;) {
		pthread_mutex_lock(&mu);
		while (fetched == g_submitted && !g_reader_done)
			pthread_cond_wait(&cv, &mu);
		int more = fetched < g_submitted, err = g_reader_err;
		pthread_mutex_unlock(&mu);
		if (!more) {
			if (err) {
				fprintf(stderr, err == 3 ? "not BGZF or corrupted\n" : err == 2 ? "read error\n" : "inflate: submit failed\n");
				ret = err == 3 ? -1 : 1;
			}
			break;
		}
		const uint8_t *data;
		size_t nbytes;
		int r = hipdeflate_unpipe_result(g_unpipes[fetched % g_npipe], &data, &nbytes);
		if (r) {
			fprintf(stderr, "inflate %d\n", r);                          /* applet/7bgzf.c:350-353 */
			ret = 1;
			break;
		}
		fetched++;
		if (write_all(data, nbytes)) {
			fprintf(stderr, "write error\n");
			ret = 1;
			break;
		}
	}
	if (ret)
		_exit(ret < 0 ? 255 : 1);
	pthread_join(rd, NULL);
	for (int k = 0; k < g_npipe; k++)
		hipdeflate_unpipe_close(g_unpipes[k]);
	fprintf(stderr, "%d done.\n", g_members);
	return 0;
}

__attribute__((constructor)) static void stamp_ctor(void) { g_t_ctor = now_s(); }

int main(int argc, char **argv)
{
	g_t_main = now_s();
	int level = -1, decode = 0, bsize = 512, nthreads = 8;
	const char *in_path = NULL, *out_path = NULL;
	for (int i = 1; i < argc; i++) {
		const char *a = argv[i];
		if (!strcmp(a, "-d") || !strcmp(a, "--decompress"))
			decode = 1;
		else if (!strncmp(a, "-G", 2) || !strncmp(a, "-l", 2))
			level = a[2] ? atoi(a + 2) : 1;
		else if (!strncmp(a, "--hip", 5))
			level = a[5] == '=' ? atoi(a + 6) : 1;
		else if (!strcmp(a, "-M") || !strcmp(a, "--migz"))
			g_frame = HD_FRAME_MIGZ;
		else if (!strncmp(a, "-b", 2))
			bsize = atoi(a + 2);
		else if (!strcmp(a, "--index") && i + 1 < argc)
			g_index_path = argv[++i];
		else if (!strcmp(a, "-i") && i + 1 < argc)
			in_path = argv[++i];
		else if (!strcmp(a, "-o") && i + 1 < argc)
			out_path = argv[++i];
		else if (!strncmp(a, "-@", 2))
			nthreads = a[2] ? atoi(a + 2) : 8;              /* I/O threads of the file-to-file path */
		else if (!strncmp(a, "-g", 2) && a[2] >= '0' && a[2] <= '9')
			g_npipe = atoi(a + 2);                          /* devices */
		else if (!strcmp(a, "-c"))
			;                                               /* accepted and ignored, as the reference's -c */
		else {
			fprintf(stderr, "usage: %s -G<level> < dec.bin > enc.bgz   or   -d < enc.bgz > dec.bin   [-M -b<KiB>]\n",
				argv[0]);
			return 1;
		}
	}
	if (!decode && level < 0) {
		fprintf(stderr, "usage: %s -G<level> < dec.bin > enc.bgz   or   -d < enc.bgz > dec.bin   [-M -b<KiB>]\n", argv[0]);
		return 1;
	}
	if (g_frame == HD_FRAME_MIGZ) {
		g_block = (size_t)bsize * 1024;
	}
	g_timing = getenv("HD7BGZF_TIMING") != NULL;
	if (g_npipe < 1 || g_npipe > HD_MAX_G) {
		fprintf(stderr, "-g: 1..%d devices\n", HD_MAX_G);
		return 1;
	}
	if (g_npipe > 1 && !(getenv("HIPDEFLATE_DEVICES") && *getenv("HIPDEFLATE_DEVICES"))) {
		int list[HD_MAX_G];                                   /* -g N without a list: devices 0 .. N-1 */
		for (int k = 0; k < g_npipe; k++)
			list[k] = k;
		if (hipdeflate_init_devices(list, g_npipe))
			return 1;
	} else if (hipdeflate_init(-1)) {                             /* (-1: HIPDEFLATE_DEVICES if set, else one device) */
		return 1;
	}
	if (hipdeflate_device_count() < g_npipe) {
		fprintf(stderr, "-g %d: the device list (HIPDEFLATE_DEVICES) has %d entries\n", g_npipe, hipdeflate_device_count());
		return 1;
	}
	struct timeval t0, t1;
	gettimeofday(&t0, NULL);
	int ret;
	if (decode) {
		/* -d -i FILE -o FILE: the filter on the two files (the decoder reads and writes in stream order) */
		if (in_path) {
			const int fd = open(in_path, O_RDONLY);
			if (fd < 0 || dup2(fd, 0) < 0) {
				fprintf(stderr, "cannot open %s\n", in_path);
				return 1;
			}
			close(fd);
		}
		if (out_path) {
			const int fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
			if (fd < 0 || dup2(fd, 1) < 0) {
				fprintf(stderr, "cannot open %s\n", out_path);
				return 1;
			}
			close(fd);
		}
		ret = do_decompress();
	} else if (in_path && out_path) {
		struct stat sb;
		g_fd_in = open(in_path, O_RDONLY);
		g_fd_out = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
		if (g_fd_in < 0 || g_fd_out < 0 || fstat(g_fd_in, &sb)) {
			fprintf(stderr, "cannot open %s / %s\n", in_path, out_path);
			return 1;
		}
		g_in_size = sb.st_size;
		if (nthreads < 1)
			nthreads = 1;
		if (nthreads > 64)
			nthreads = 64;
		{
			/* N readers + N writers beside the HIP runtime's own threads: more runnable threads than CPUs made -@8 and
			 * -@16 slower than -@4 on a 16-CPU box (round 3).  The I/O threads stay within the CPUs this process may use. */
			cpu_set_t set;
			int ncpu = 0;
			if (!sched_getaffinity(0, sizeof(set), &set))
				ncpu = CPU_COUNT(&set);
			/* (... and within the cgroup's quota: a 16-CPU container on a 256-core host reports 256 through the affinity mask --
			 * round 5: -@16 was thirty-two I/O threads there) */
			FILE *cf = fopen("/sys/fs/cgroup/cpu.max", "r");
			if (cf) {
				long long q = 0, per = 0;
				if (fscanf(cf, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0 && (int)((q + per - 1) / per) < ncpu)
					ncpu = (int)((q + per - 1) / per);
				fclose(cf);
			}
			const int most = ncpu > 5 ? (ncpu - 2) / 2 : 2;
			if (nthreads > most)
				nthreads = most;
		}
		fprintf(stderr, "compression level = %d (hip)\n", level);
		ret = do_compress_files(level, nthreads);
		close(g_fd_in);
		close(g_fd_out);
	} else {
		fprintf(stderr, "compression level = %d (hip)\n", level);     /* applet/7bgzf.c:502-524 */
		ret = do_compress(level);
	}
	fflush(stdout);
	gettimeofday(&t1, NULL);
	fprintf(stderr, "ellapsed time: %f sec\n", (t1.tv_sec - t0.tv_sec) + (t1.tv_usec - t0.tv_usec) * 1e-6);
	if (g_timing)
		fprintf(stderr, "timing: main() began %.3f s after the process' first instruction-side stamp, the job ended at %.3f s\n",
			g_t_main - g_t_ctor, now_s() - g_t_ctor);
	hipdeflate_shutdown();
	if (g_timing)
		fprintf(stderr, "timing: hipdeflate_shutdown() returned at %.3f s (what follows is the HIP runtime's own exit)\n", now_s() - g_t_ctor);
	return ret;
}
