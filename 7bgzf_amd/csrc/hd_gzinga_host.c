/*
 * hd_gzinga_host.c -- hd7gzinga: applet/7gzinga.c (_compress :77-197, _decompress
 * :199-305) over libhipdeflate.so.  GZinga is a seekable / splittable gzip: every
 * 100 KiB block is its own member whose header carries an (empty) comment, and a
 * last member with no data holds the index as ITS comment: "k:<end offset of block
 * k>;" for every block.
 *
 *     hd7gzinga -G<level> < dec.bin > enc.gz
 *     hd7gzinga -d enc.gz > dec.bin
 *
 * What changed, and why: blocks go to the device in batches (HD_FRAME_RAW; CRC-32 of
 * each block comes back with it, so there is no fcrc32 pass on the host, :176) and
 * the reader hands whole runs of members to one batched inflate, checking CRC-32
 * and ISIZE of every member (the reference checks neither).  The reference reads
 * the index from the last 32 KiB of the file only (:212-216), which bounds ITS
 * reader to about 2,500 blocks; this reader widens the search until it finds the
 * index member.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "hipdeflate.h"
#include "hd_host_util.h"

#define GZ_BLOCK (100u * 1024u)
#define GZ_BATCH 1024
#define GZ_MAX_ISIZE (64u << 20)      /* a member claiming more than this is not ours nor the reference's */

static const unsigned char gz_header[9] = { 0x1f, 0x8b, 0x08, 0x10, 0, 0, 0, 0, 0x00 };

static size_t read_full(FILE *f, unsigned char *buf, size_t want)
{
	size_t got = 0;
	while (got < want) {
		size_t r = fread(buf + got, 1, want - got, f);
		if (!r)
			break;
		got += r;
	}
	return got;
}

static int gz_compress(FILE *in, FILE *out, int level)
{
	const size_t stride = up16(GZ_BLOCK + 5 * 3 + 32);
	unsigned char *ibuf = malloc((size_t)GZ_BATCH * GZ_BLOCK + 16);
	unsigned char *obuf = malloc((size_t)GZ_BATCH * stride + 16);
	uint64_t *off = malloc(sizeof(uint64_t) * GZ_BATCH);
	uint32_t *len = malloc(sizeof(uint32_t) * GZ_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * GZ_BATCH);
	uint32_t *crc = malloc(sizeof(uint32_t) * GZ_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * GZ_BATCH);
	uint64_t *ends = NULL;
	size_t nends = 0, cap_ends = 0;
	if (!ibuf || !obuf || !off || !len || !olen || !crc || !st) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	uint64_t total_size = 0;
	int ret = 0;
	for (;;) {
		const size_t got = read_full(in, ibuf, (size_t)GZ_BATCH * GZ_BLOCK);
		if (!got)
			break;
		const uint32_t n = (uint32_t)((got + GZ_BLOCK - 1) / GZ_BLOCK);
		for (uint32_t i = 0; i < n; i++) {
			off[i] = (uint64_t)i * GZ_BLOCK;
			len[i] = got - off[i] < GZ_BLOCK ? (uint32_t)(got - off[i]) : GZ_BLOCK;
		}
		int r = hipdeflate_batch_deflate(ibuf, off, len, n, level, HD_FRAME_RAW, obuf, stride, (uint32_t)stride, olen, crc, st);
		if (r) {
			fprintf(stderr, "hip_deflate %d\n", r);
			ret = 1;
			break;
		}
		if (nends + n > cap_ends) {
			cap_ends = (nends + n) * 2;
			ends = realloc(ends, cap_ends * sizeof(uint64_t));
			if (!ends) {
				fprintf(stderr, "out of memory\n");
				ret = 2;
				break;
			}
		}
		for (uint32_t i = 0; i < n; i++) {
			if (st[i]) {
				fprintf(stderr, "hip_deflate %d\n", st[i]);
				ret = 1;
				break;
			}
			unsigned char t[11];
			memcpy(t, gz_header, 9);
			t[9] = 0xff, t[10] = 0x00;                /* OS = unknown, then the empty comment */
			fwrite(t, 1, 11, out);
			fwrite(obuf + (size_t)i * stride, 1, olen[i], out);
			wr32(t, crc[i]);
			wr32(t + 4, len[i]);
			fwrite(t, 1, 8, out);
			total_size += 11 + (uint64_t)olen[i] + 8;
			ends[nends++] = total_size;
		}
		if (ret)
			break;
		fprintf(stderr, "%zu\r", nends);
		if (got < (size_t)GZ_BATCH * GZ_BLOCK)
			break;
	}
	if (!ret) {
		unsigned char t[11] = { 0x1f, 0x8b, 0x08, 0x10, 0, 0, 0, 0, 0x00, 0xff };
		fwrite(t, 1, 10, out);
		for (size_t k = 0; k < nends; k++)
			fprintf(out, "%zu:%llu;", k, (unsigned long long)ends[k]);
		memset(t, 0, sizeof(t));
		t[1] = 0x03;                                          /* NUL ends the comment; 03 00 = empty final block; CRC 0, ISIZE 0 */
		fwrite(t, 1, 11, out);
		fprintf(stderr, "%zu done.\n", nends);
		if (fflush(out) || ferror(out)) {
			fprintf(stderr, "write error\n");
			ret = 2;
		}
	}
	free(ibuf), free(obuf), free(off), free(len), free(olen), free(crc), free(st), free(ends);
	return ret;
}

/* read_gz_header_generic's job: length of a gzip member header, 0 if it is none */
static size_t gz_member_header(const unsigned char *d, size_t size)
{
	if (size < 10 || d[0] != 0x1f || d[1] != 0x8b || d[2] != 8 || (d[3] & 0xe0))
		return 0;
	size_t n = 10;
	if (d[3] & 4) {
		if (size < n + 2)
			return 0;
		n += 2 + rd16(d + n);
	}
	if (d[3] & 8) {
		while (n < size && d[n])
			n++;
		n++;
	}
	if (d[3] & 16) {
		while (n < size && d[n])
			n++;
		n++;
	}
	if (d[3] & 2)
		n += 2;
	return n <= size ? n : 0;
}

static int gz_decompress(FILE *in, FILE *out)
{
	const long long fsize = file_size(in);
	if (fsize < 21) {
		fprintf(stderr, "not GZinga or corrupted\n");
		return 1;
	}
	/* the index member is the last place the 9 header bytes occur (:217-227) */
	unsigned char *foot = NULL;
	size_t foot_len = 0;
	long long index_at = -1;
	for (size_t span = 32 * 1024;; span *= 4) {
		if ((long long)span > fsize)
			span = (size_t)fsize;
		free(foot);
		foot = malloc(span + 1);
		if (!foot || fseeko(in, fsize - (long long)span, SEEK_SET) || read_full(in, foot, span) != span) {
			fprintf(stderr, "cannot read the index\n");
			free(foot);
			return 1;
		}
		foot_len = span;
		foot[span] = 0;
		/* an index is text: it cannot itself contain the header bytes, so the last occurrence whose comment
		 * runs to the file's final 11 bytes is the index member */
		for (size_t k = span >= 9 ? span - 9 + 1 : 0; k-- > 0;) {
			if (!memcmp(foot + k, gz_header, 9)) {
				const size_t text = k + 10;
				const size_t nul = text + strlen((const char *)foot + text);
				if (nul + 11 == span)
					index_at = fsize - (long long)span + (long long)k;
				break;
			}
		}
		if (index_at >= 0 || (long long)span == fsize)
			break;
	}
	if (index_at < 0) {
		fprintf(stderr, "not GZinga or corrupted\n");
		free(foot);
		return 1;
	}
	/* "k:<end>;" ... */
	const char *p = (const char *)foot + (size_t)(index_at - (fsize - (long long)foot_len)) + 10;
	uint64_t *lst = malloc(sizeof(uint64_t) * 2);
	size_t nblk = 0, cap_lst = 2;
	int ret = 0;
	if (lst)
		lst[0] = 0;
	while (lst && *p) {
		char *e;
		(void)strtoull(p, &e, 10);
		if (e == p || *e != ':') {
			ret = 1;
			break;
		}
		p = e + 1;
		const uint64_t v = strtoull(p, &e, 10);
		if (e == p || *e != ';' || v < lst[nblk] + 19 || v > (uint64_t)index_at) {
			ret = 1;
			break;
		}
		p = e + 1;
		if (nblk + 2 > cap_lst) {
			cap_lst *= 2;
			lst = realloc(lst, sizeof(uint64_t) * cap_lst);
			if (!lst)
				break;
		}
		lst[++nblk] = v;
	}
	free(foot);
	if (!lst || ret || (nblk ? lst[nblk] : 0) != (uint64_t)index_at) {
		fprintf(stderr, "corrupted index\n");
		free(lst);
		return 1;
	}
	unsigned char *ibuf = NULL, *obuf = NULL;
	size_t icap = 0, ocap = 0;
	uint64_t *ioff = malloc(sizeof(uint64_t) * GZ_BATCH), *ooff = malloc(sizeof(uint64_t) * GZ_BATCH);
	uint32_t *ilen = malloc(sizeof(uint32_t) * GZ_BATCH), *cap = malloc(sizeof(uint32_t) * GZ_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * GZ_BATCH), *crc = malloc(sizeof(uint32_t) * GZ_BATCH);
	uint32_t *want_crc = malloc(sizeof(uint32_t) * GZ_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * GZ_BATCH);
	if (!ioff || !ooff || !ilen || !cap || !olen || !crc || !want_crc || !st) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	for (size_t c = 0; c < nblk && !ret; c += GZ_BATCH) {
		const uint32_t m = (uint32_t)(nblk - c < GZ_BATCH ? nblk - c : GZ_BATCH);
		const size_t itotal = (size_t)(lst[c + m] - lst[c]);
		if (itotal + 16 > icap) {
			free(ibuf);
			ibuf = malloc(icap = itotal + 16);
		}
		if (!ibuf || fseeko(in, (long long)lst[c], SEEK_SET) || read_full(in, ibuf, itotal) != itotal) {
			fprintf(stderr, "file truncated\n");
			ret = 1;
			break;
		}
		size_t ototal = 0;
		for (uint32_t i = 0; i < m; i++) {
			const size_t a = (size_t)(lst[c + i] - lst[c]), mlen = (size_t)(lst[c + i + 1] - lst[c + i]);
			const size_t n = gz_member_header(ibuf + a, mlen);
			if (!n || n + 8 > mlen || rd32(ibuf + a + mlen - 4) > GZ_MAX_ISIZE) {
				fprintf(stderr, "corrupted\n");
				ret = 1;
				break;
			}
			ioff[i] = a + n;
			ilen[i] = (uint32_t)(mlen - n - 8);
			cap[i] = rd32(ibuf + a + mlen - 4);
			want_crc[i] = rd32(ibuf + a + mlen - 8);
			ooff[i] = ototal;
			ototal += up16(cap[i]);
		}
		if (ret)
			break;
		if (ototal + 16 > ocap) {
			free(obuf);
			obuf = malloc(ocap = ototal + 16);
			if (!obuf) {
				fprintf(stderr, "out of memory\n");
				ret = 2;
				break;
			}
		}
		int r = hipdeflate_batch_inflate(ibuf, ioff, ilen, m, obuf, ooff, cap, olen, crc, st);
		if (r) {
			fprintf(stderr, "inflate %d\n", r);
			ret = 1;
			break;
		}
		for (uint32_t i = 0; i < m; i++) {
			if (st[i]) {
				fprintf(stderr, "inflate %d\n", st[i]);
				ret = 1;
				break;
			}
			if (olen[i] != cap[i] || crc[i] != want_crc[i]) {
				fprintf(stderr, "crc32 / size mismatch\n");
				ret = 1;
				break;
			}
			fwrite(obuf + ooff[i], 1, olen[i], out);
		}
		fprintf(stderr, "%zu\r", c + m);
	}
	if (!ret) {
		fprintf(stderr, "%zu done.\n", nblk);
		if (fflush(out) || ferror(out)) {
			fprintf(stderr, "write error\n");
			ret = 2;
		}
	}
	free(lst), free(ibuf), free(obuf), free(ioff), free(ooff), free(ilen), free(cap), free(olen), free(crc), free(want_crc),
		free(st);
	return ret;
}

int main(int argc, char **argv)
{
	int level = -1, decode = 0, bad = 0;
	const char *name = NULL;
	for (int i = 1; i < argc; i++) {
		const char *a = argv[i];
		if (a[0] == '-' && a[1]) {
			for (const char *p = a + 1; *p; p++) {
				if (*p == 'd')
					decode = 1;
				else if (*p == 'c')
					;
				else if (*p == '@')
					break;                          /* -@<threads>: accepted and ignored */
				else if (*p == 'G' || *p == 'l') {
					level = p[1] ? atoi(p + 1) : 1;
					break;
				} else {
					bad = 1;
					break;
				}
			}
		} else if (!name) {
			name = a;
		} else {
			bad = 1;
		}
	}
	if (bad || (decode && (!name || level >= 0)) || (!decode && (name || level < 0 || level > 9)) ||
	    (!decode && (isatty(0) || isatty(1)))) {
		fprintf(stderr, "usage: %s -G<level> < dec.bin > enc.gz   or   -d enc.gz > dec.bin\n", argv[0]);
		return 1;
	}
	int r = hipdeflate_init(-1);
	if (r) {
		fprintf(stderr, "hipdeflate: no usable device (%d): %s\n", r, hipdeflate_version());
		return 4;
	}
	const double t0 = now_s();
	int ret;
	if (decode) {
		FILE *in = fopen(name, "rb");
		if (!in) {
			fprintf(stderr, "failed to open %s\n", name);
			return 2;
		}
		ret = gz_decompress(in, stdout);
		fclose(in);
	} else {
		fprintf(stderr, "compression level = %d (hip)\n", level);
		ret = gz_compress(stdin, stdout, level);
	}
	fprintf(stderr, "ellapsed time: %.3f sec\n", now_s() - t0);
	hipdeflate_shutdown();
	return ret;
}
