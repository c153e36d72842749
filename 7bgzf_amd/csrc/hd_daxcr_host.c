/*
 * hd_daxcr_host.c -- hd7daxcr: applet/7daxcr.c (_compress :72-176, _decompress
 * :178-239) over libhipdeflate.so.  DAX: a 32-byte header, a table of 32-bit file
 * offsets and one of 16-bit sizes, then every 8192-byte frame as its own RFC 1950
 * (zlib) stream -- the container that goes through zlibutil_buffer_code's rfc1950
 * wrapper (lib/zlibutil.c:374-397) in the reference.
 *
 *     hd7daxcr -G<level> dec.iso enc.dax
 *     hd7daxcr -d < enc.dax > dec.iso
 *
 * What changed, and why: frames go to the device 32,768 at a time and come back as
 * finished zlib members (HD_FRAME_ZLIB: 78 da, the stream, Adler-32 computed on the
 * device).  The reader inflates the members' payloads in batches; non-compressed
 * areas (which neither writer makes) are honoured.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "hipdeflate.h"
#include "hd_host_util.h"

#define DX_BLOCK 8192u
#define DX_BATCH 32768u

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

static int dx_compress(FILE *in, FILE *out, int level)
{
	const long long total = file_size(in);
	if (total < 0 || total >= (1ll << 32)) {
		fprintf(stderr, total < 0 ? "cannot stat the input\n" : "input too large for a DAX header\n");
		return 2;
	}
	const uint32_t nblk = (uint32_t)((total + DX_BLOCK - 1) / DX_BLOCK);
	const size_t stride = (size_t)hipdeflate_bound(DX_BLOCK, level);
	unsigned char hdr[32] = { 'D', 'A', 'X', 0 };
	wr32(hdr + 4, (uint32_t)total);
	wr32(hdr + 8, 1);
	unsigned char *index = calloc(6, (size_t)nblk + 1);
	unsigned char *ibuf = malloc((size_t)DX_BATCH * DX_BLOCK + 16);
	unsigned char *obuf = malloc((size_t)DX_BATCH * stride + 16);
	uint64_t *off = malloc(sizeof(uint64_t) * DX_BATCH);
	uint32_t *len = malloc(sizeof(uint32_t) * DX_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * DX_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * DX_BATCH);
	if (!index || !ibuf || !obuf || !off || !len || !olen || !st) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	unsigned char *sizes = index + 4 * (size_t)nblk;
	fwrite(hdr, 1, 32, out);
	fwrite(index, 1, 6 * (size_t)nblk, out);
	uint64_t pos = 32 + 6 * (uint64_t)nblk;
	long long left = total;
	int ret = 0;
	for (uint32_t c = 0; c < nblk && !ret; c += DX_BATCH) {
		const uint32_t n = nblk - c < DX_BATCH ? nblk - c : DX_BATCH;
		const size_t want = left < (long long)n * DX_BLOCK ? (size_t)left : (size_t)n * DX_BLOCK;
		if (fread(ibuf, 1, want, in) != want) {
			fprintf(stderr, "short read\n");
			ret = 2;
			break;
		}
		for (uint32_t i = 0; i < n; i++) {
			off[i] = (uint64_t)i * DX_BLOCK;
			len[i] = want - (size_t)off[i] < DX_BLOCK ? (uint32_t)(want - (size_t)off[i]) : DX_BLOCK;
		}
		int r = hipdeflate_batch_deflate(ibuf, off, len, n, level, HD_FRAME_ZLIB, obuf, stride, (uint32_t)stride, olen, NULL, st);
		if (r) {
			fprintf(stderr, "hip_deflate %d\n", r);
			ret = 1;
			break;
		}
		for (uint32_t i = 0; i < n; i++) {
			if (st[i] || olen[i] > 0xffff) {
				fprintf(stderr, "hip_deflate %d\n", st[i] ? st[i] : 1);
				ret = 1;
				break;
			}
			if (pos >= (1ull << 32)) {
				fprintf(stderr, "output too large for 32-bit DAX offsets\n");
				ret = 2;
				break;
			}
			wr32(index + 4 * (size_t)(c + i), (uint32_t)pos);
			wr16(sizes + 2 * (size_t)(c + i), olen[i]);
			fwrite(obuf + (size_t)i * stride, 1, olen[i], out);
			pos += olen[i];
		}
		left -= (long long)want;
		fprintf(stderr, "%u / %u\r", c + n, nblk);
	}
	if (!ret) {
		fseeko(out, 32, SEEK_SET);
		fwrite(index, 1, 6 * (size_t)nblk, out);
		fprintf(stderr, "%u / %u done.\n", nblk, nblk);
		if (fflush(out) || ferror(out)) {
			fprintf(stderr, "write error\n");
			ret = 2;
		}
	}
	free(index), free(ibuf), free(obuf), free(off), free(len), free(olen), free(st);
	return ret;
}

static int dx_decompress(FILE *in, FILE *out)
{
	unsigned char hdr[32];
	if (read_full(in, hdr, 32) != 32 || memcmp(hdr, "DAX\0", 4)) {
		fprintf(stderr, "not DAX\n");
		return 1;
	}
	const uint32_t total = rd32(hdr + 4), nnc = rd32(hdr + 12);
	const uint32_t nblk = (uint32_t)(((uint64_t)total + DX_BLOCK - 1) / DX_BLOCK);
	if (nnc > nblk) {
		fprintf(stderr, "not DAX\n");
		return 1;
	}
	unsigned char *index = malloc(6 * (size_t)nblk + 8 * (size_t)nnc + 16);
	if (!index || read_full(in, index, 6 * (size_t)nblk + 8 * (size_t)nnc) != 6 * (size_t)nblk + 8 * (size_t)nnc) {
		fprintf(stderr, "unexpected end of file\n");
		return 1;
	}
	const unsigned char *sizes = index + 4 * (size_t)nblk, *nc = sizes + 2 * (size_t)nblk;
	/* frames of the non-compressed areas: plain 8192-byte frames in the file (:194-203) */
	unsigned char *plain = calloc(1, (size_t)nblk + 1);
	unsigned char *ibuf = malloc((size_t)DX_BATCH * (DX_BLOCK + 64) + 16), *obuf = malloc((size_t)DX_BATCH * DX_BLOCK + 16);
	uint64_t *ioff = malloc(sizeof(uint64_t) * DX_BATCH), *ooff = malloc(sizeof(uint64_t) * DX_BATCH);
	uint32_t *ilen = malloc(sizeof(uint32_t) * DX_BATCH), *cap = malloc(sizeof(uint32_t) * DX_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * DX_BATCH), *map = malloc(sizeof(uint32_t) * DX_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * DX_BATCH);
	if (!plain || !ibuf || !obuf || !ioff || !ooff || !ilen || !cap || !olen || !map || !st) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	for (uint32_t k = 0; k < nnc; k++) {
		const uint32_t first = rd32(nc + 8 * (size_t)k), cnt = rd32(nc + 8 * (size_t)k + 4);
		for (uint32_t j = 0; j < cnt && (uint64_t)first + j < nblk; j++)
			plain[first + j] = 1;
	}
	uint64_t produced = 0;
	int ret = 0;
	for (uint32_t c = 0; c < nblk && !ret; c += DX_BATCH) {
		const uint32_t m = nblk - c < DX_BATCH ? nblk - c : DX_BATCH;
		size_t itotal = 0;
		uint32_t nz = 0;
		for (uint32_t i = 0; i < m; i++) {
			const uint32_t want = (uint64_t)(c + i + 1) * DX_BLOCK <= total ? DX_BLOCK : (uint32_t)(total - (uint64_t)(c + i) * DX_BLOCK);
			const uint32_t sz = plain[c + i] ? DX_BLOCK : rd16(sizes + 2 * (size_t)(c + i));
			if (sz > DX_BLOCK + 64 || (!plain[c + i] && sz < 6)) {
				ret = 1;
				break;
			}
			if (!plain[c + i]) {
				ioff[nz] = itotal + 2;                 /* behind the two zlib header bytes */
				ilen[nz] = sz - 2;                     /* the Adler-32 rides along as trailing bytes */
				ooff[nz] = (uint64_t)i * DX_BLOCK;
				cap[nz] = want;
				map[nz++] = i;
			}
			itotal += sz;
		}
		if (ret) {
			fprintf(stderr, "corrupted size table\n");
			break;
		}
		if (read_full(in, ibuf, itotal) != itotal) {
			fprintf(stderr, "unexpected end of file\n");
			ret = 1;
			break;
		}
		size_t at = 0;
		for (uint32_t i = 0; i < m; i++) {
			if (plain[c + i]) {
				memcpy(obuf + (size_t)i * DX_BLOCK, ibuf + at, DX_BLOCK);
				at += DX_BLOCK;
			} else {
				if ((ibuf[at] & 0x0f) != 8 || ((ibuf[at] << 8) | ibuf[at + 1]) % 31) {
					fprintf(stderr, "frame %u is not a zlib stream\n", c + i);
					ret = 1;
					break;
				}
				at += rd16(sizes + 2 * (size_t)(c + i));
			}
		}
		if (ret)
			break;
		if (nz) {
			int r = hipdeflate_batch_inflate(ibuf, ioff, ilen, nz, obuf, ooff, cap, olen, NULL, st);
			if (r) {
				fprintf(stderr, "inflate %d\n", r);
				ret = 1;
				break;
			}
		}
		for (uint32_t k = 0; k < nz; k++) {
			if (st[k] || olen[k] != cap[k]) {
				fprintf(stderr, "inflate %d\n", st[k] ? st[k] : 1);
				ret = 1;
				break;
			}
		}
		if (ret)
			break;
		const uint64_t bytes = produced + (uint64_t)m * DX_BLOCK <= total ? (uint64_t)m * DX_BLOCK : total - produced;
		fwrite(obuf, 1, (size_t)bytes, out);
		produced += bytes;
		fprintf(stderr, "%u / %u\r", c + m, nblk);
	}
	if (!ret) {
		fprintf(stderr, "%u / %u done.\n", nblk, nblk);
		if (fflush(out) || ferror(out)) {
			fprintf(stderr, "write error\n");
			ret = 2;
		}
	}
	free(index), free(plain), free(ibuf), free(obuf), free(ioff), free(ooff), free(ilen), free(cap), free(olen), free(map), free(st);
	return ret;
}

int main(int argc, char **argv)
{
	int level = -1, decode = 0, bad = 0, nn = 0;
	const char *names[2] = { NULL, NULL };
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
		} else if (nn < 2) {
			names[nn++] = a;
		} else {
			bad = 1;
		}
	}
	if (bad || (decode && (nn || level >= 0)) || (!decode && (nn != 2 || level < 0 || level > 9)) ||
	    (decode && (isatty(0) || isatty(1)))) {
		fprintf(stderr, "usage: %s -G<level> dec.iso enc.dax   or   -d < enc.dax > dec.iso\n", argv[0]);
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
		ret = dx_decompress(stdin, stdout);
	} else {
		FILE *in = fopen(names[0], "rb");
		if (!in) {
			fprintf(stderr, "failed to open %s\n", names[0]);
			return 2;
		}
		FILE *out = fopen(names[1], "wb");
		if (!out) {
			fprintf(stderr, "failed to open %s\n", names[1]);
			fclose(in);
			return 3;
		}
		fprintf(stderr, "compression level = %d (hip)\n", level);
		ret = dx_compress(in, out, level);
		fclose(in);
		if (fclose(out) && !ret)
			ret = 2;
	}
	fprintf(stderr, "ellapsed time: %.3f sec\n", now_s() - t0);
	hipdeflate_shutdown();
	return ret;
}
