/*
 * hd_ciso_host.c -- hd7ciso: applet/7ciso.c (_compress :81-208, _decompress
 * :210-293) over libhipdeflate.so.  CISO: a 24-byte header, a table of
 * (sectors + 1) 32-bit file offsets (bit 31 = the sector is stored as it is), then
 * every 2048-byte sector as its own raw DEFLATE stream.  The tiny-block end of the
 * path: a 1 GiB image is 524,288 independent blocks.
 *
 *     hd7ciso -G<level> [-t<percent>] dec.iso enc.cso
 *     hd7ciso -d < enc.cso > dec.iso
 *
 * What changed, and why: sectors go to the device 65,536 at a time in one call
 * (the reference: one pthread per sector); a sector whose stream is longer than
 * threshold % of 2048 is written plain, as there (:188-193).
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "hipdeflate.h"
#include "hd_host_util.h"

#define CS_BLOCK 2048u
#define CS_BATCH 65536u

static int cs_compress(FILE *in, FILE *out, int level, int threshold)
{
	const long long total = file_size(in);
	if (total < 0 || total >= (1ll << 31)) {
		fprintf(stderr, total < 0 ? "cannot stat the input\n" : "input too large for 31-bit CISO offsets\n");
		return 2;
	}
	const uint32_t nblk = (uint32_t)((total + CS_BLOCK - 1) / CS_BLOCK);
	const size_t stride = up16(CS_BLOCK + 5 + 32);
	unsigned char hdr[24] = { 'C', 'I', 'S', 'O' };
	wr32(hdr + 4, 24);
	wr32(hdr + 8, (uint32_t)total);
	wr32(hdr + 12, (uint32_t)((uint64_t)total >> 32));
	wr32(hdr + 16, CS_BLOCK);
	hdr[20] = 1;                                     /* ver; align = 0 */
	unsigned char *index = calloc(4, (size_t)nblk + 1);
	unsigned char *ibuf = malloc((size_t)CS_BATCH * CS_BLOCK + 16);
	unsigned char *obuf = malloc((size_t)CS_BATCH * stride + 16);
	uint64_t *off = malloc(sizeof(uint64_t) * CS_BATCH);
	uint32_t *len = malloc(sizeof(uint32_t) * CS_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * CS_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * CS_BATCH);
	if (!index || !ibuf || !obuf || !off || !len || !olen || !st) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	fwrite(hdr, 1, 24, out);
	fwrite(index, 4, (size_t)nblk + 1, out);
	uint64_t pos = 24 + 4 * ((uint64_t)nblk + 1);
	long long left = total;
	int ret = 0;
	for (uint32_t c = 0; c < nblk && !ret; c += CS_BATCH) {
		const uint32_t n = nblk - c < CS_BATCH ? nblk - c : CS_BATCH;
		const size_t want = left < (long long)n * CS_BLOCK ? (size_t)left : (size_t)n * CS_BLOCK;
		if (fread(ibuf, 1, want, in) != want) {
			fprintf(stderr, "short read\n");
			ret = 2;
			break;
		}
		for (uint32_t i = 0; i < n; i++) {
			off[i] = (uint64_t)i * CS_BLOCK;
			len[i] = want - (size_t)off[i] < CS_BLOCK ? (uint32_t)(want - (size_t)off[i]) : CS_BLOCK;
		}
		int r = hipdeflate_batch_deflate(ibuf, off, len, n, level, HD_FRAME_RAW, obuf, stride, (uint32_t)stride, olen, NULL, st);
		if (r) {
			fprintf(stderr, "hip_deflate %d\n", r);
			ret = 1;
			break;
		}
		for (uint32_t i = 0; i < n; i++) {
			if (st[i]) {
				fprintf(stderr, "hip_deflate %d\n", st[i]);
				ret = 1;
				break;
			}
			if (pos >= (1ull << 31)) {
				fprintf(stderr, "output too large for 31-bit CISO offsets\n");
				ret = 2;
				break;
			}
			if (olen[i] > CS_BLOCK * (uint32_t)threshold / 100) {
				wr32(index + 4 * (size_t)(c + i), 0x80000000u | (uint32_t)pos);
				fwrite(ibuf + off[i], 1, len[i], out);
				pos += len[i];
			} else {
				wr32(index + 4 * (size_t)(c + i), (uint32_t)pos);
				fwrite(obuf + (size_t)i * stride, 1, olen[i], out);
				pos += olen[i];
			}
		}
		left -= (long long)want;
		fprintf(stderr, "%u / %u\r", c + n, nblk);
	}
	if (!ret) {
		wr32(index + 4 * (size_t)nblk, (uint32_t)pos);
		fseeko(out, 24, SEEK_SET);
		fwrite(index, 4, (size_t)nblk + 1, out);
		fprintf(stderr, "%u / %u done.\n", nblk, nblk);
		if (fflush(out) || ferror(out)) {
			fprintf(stderr, "write error\n");
			ret = 2;
		}
	}
	free(index), free(ibuf), free(obuf), free(off), free(len), free(olen), free(st);
	return ret;
}

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

static int cs_decompress(FILE *in, FILE *out)
{
	unsigned char hdr[24];
	if (read_full(in, hdr, 24) != 24 || memcmp(hdr, "CISO", 4) || (rd32(hdr + 4) && rd32(hdr + 4) != 24)) {
		fprintf(stderr, "not CISO\n");
		return 1;
	}
	const uint64_t total = rd32(hdr + 8) | ((uint64_t)rd32(hdr + 12) << 32);
	const uint32_t block = rd32(hdr + 16), align = hdr[21];
	if (!block || block > (1u << 20) || align > 8 || total / block >= (1u << 27)) {
		fprintf(stderr, "not CISO\n");
		return 1;
	}
	const uint32_t nblk = (uint32_t)((total + block - 1) / block);
	unsigned char *index = malloc(4 * ((size_t)nblk + 1));
	if (!index || read_full(in, index, 4 * ((size_t)nblk + 1)) != 4 * ((size_t)nblk + 1)) {
		fprintf(stderr, "unexpected end of file\n");
		return 1;
	}
	/* the sector size comes from the file: a batch holds at most 128 MiB of output whatever it claims (2048-byte
	 * sectors: the whole CS_BATCH; a header that says 1 MiB: 128 of them -- not a 64 GiB allocation) */
	const uint32_t batch = (uint32_t)((128u << 20) / up16(block)) < CS_BATCH ? (uint32_t)((128u << 20) / up16(block)) : CS_BATCH;
	unsigned char *ibuf = NULL, *obuf = malloc((size_t)batch * up16(block) + 16);
	size_t icap = 0;
	uint64_t *ioff = malloc(sizeof(uint64_t) * CS_BATCH), *ooff = malloc(sizeof(uint64_t) * CS_BATCH);
	uint32_t *ilen = malloc(sizeof(uint32_t) * CS_BATCH), *cap = malloc(sizeof(uint32_t) * CS_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * CS_BATCH), *map = malloc(sizeof(uint32_t) * CS_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * CS_BATCH);
	if (!obuf || !ioff || !ooff || !ilen || !cap || !olen || !map || !st) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	uint64_t at = 24 + 4 * ((uint64_t)nblk + 1);       /* stdin is a pipe: the position is counted, not sought (:216) */
	uint64_t produced = 0;
	int ret = 0;
#define CS_POS(k) ((uint64_t)(rd32(index + 4 * (size_t)(k)) & 0x7fffffffu) << align)
	for (uint32_t c = 0; c < nblk && !ret; c += batch) {
		const uint32_t m = nblk - c < batch ? nblk - c : batch;
		const uint64_t first = CS_POS(c), end = CS_POS(c + m);
		if (first < at || end < first || end - first > (uint64_t)m * (block + 64)) {
			fprintf(stderr, "corrupted index\n");
			ret = 1;
			break;
		}
		const size_t span = (size_t)(end - at);             /* bytes between sectors are skipped, as there (:232) */
		if (span + 16 > icap) {
			free(ibuf);
			ibuf = malloc(icap = span + 16);
			if (!ibuf) {
				fprintf(stderr, "out of memory\n");
				ret = 2;
				break;
			}
		}
		if (read_full(in, ibuf, span) != span) {
			fprintf(stderr, "unexpected end of file\n");
			ret = 1;
			break;
		}
		uint32_t nz = 0;
		for (uint32_t i = 0; i < m && !ret; i++) {
			const uint64_t a = CS_POS(c + i), b = CS_POS(c + i + 1);
			const uint32_t want = produced + (uint64_t)(i + 1) * block <= total ? block : (uint32_t)(total - produced - (uint64_t)i * block);
			if (a < at || b < a || b > end) {
				ret = 1;
			} else if (rd32(index + 4 * (size_t)(c + i)) & 0x80000000u) {
				if (b - a < want)
					ret = 1;
				else
					memcpy(obuf + (size_t)i * up16(block), ibuf + (size_t)(a - at), want);
			} else {
				ioff[nz] = a - at;
				ilen[nz] = (uint32_t)(b - a);
				ooff[nz] = (uint64_t)i * up16(block);
				cap[nz] = block;
				map[nz++] = i;
			}
		}
		if (ret) {
			fprintf(stderr, "corrupted index\n");
			break;
		}
		if (nz) {
			int r = hipdeflate_batch_inflate(ibuf, ioff, ilen, nz, obuf, ooff, cap, olen, NULL, st);
			if (r) {
				fprintf(stderr, "inflate %d\n", r);
				ret = 1;
				break;
			}
		}
		for (uint32_t k = 0; k < nz; k++) {
			const uint32_t i = map[k];
			const uint32_t want = produced + (uint64_t)(i + 1) * block <= total ? block : (uint32_t)(total - produced - (uint64_t)i * block);
			if (st[k] || olen[k] != want) {
				fprintf(stderr, "inflate %d\n", st[k] ? st[k] : 1);
				ret = 1;
				break;
			}
		}
		if (ret)
			break;
		const uint64_t bytes = produced + (uint64_t)m * block <= total ? (uint64_t)m * block : total - produced;
		if (up16(block) == block) {
			fwrite(obuf, 1, (size_t)bytes, out);
		} else {
			for (uint32_t i = 0; i < m; i++) {
				const uint64_t o = (uint64_t)i * block;
				if (o < bytes)
					fwrite(obuf + (size_t)i * up16(block), 1, bytes - o < block ? (size_t)(bytes - o) : block, out);
			}
		}
		produced += bytes;
		at = end;
		fprintf(stderr, "%u / %u\r", c + m, nblk);
	}
	if (!ret) {
		fprintf(stderr, "%u / %u done.\n", nblk, nblk);
		if (fflush(out) || ferror(out)) {
			fprintf(stderr, "write error\n");
			ret = 2;
		}
	}
	free(index), free(ibuf), free(obuf), free(ioff), free(ooff), free(ilen), free(cap), free(olen), free(map), free(st);
	return ret;
}

int main(int argc, char **argv)
{
	int level = -1, decode = 0, threshold = 100, bad = 0, nn = 0;
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
				} else if (*p == 't') {
					threshold = atoi(p + 1);
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
		fprintf(stderr, "usage: %s -G<level> [-t<percent>] dec.iso enc.cso   or   -d < enc.cso > dec.iso\n", argv[0]);
		return 1;
	}
	if (threshold < 10)
		threshold = 10;
	if (threshold > 100)
		threshold = 100;
	int r = hipdeflate_init(-1);
	if (r) {
		fprintf(stderr, "hipdeflate: no usable device (%d): %s\n", r, hipdeflate_version());
		return 4;
	}
	const double t0 = now_s();
	int ret;
	if (decode) {
		ret = cs_decompress(stdin, stdout);
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
		ret = cs_compress(in, out, level, threshold);
		fclose(in);
		if (fclose(out) && !ret)
			ret = 2;
	}
	fprintf(stderr, "ellapsed time: %.3f sec\n", now_s() - t0);
	hipdeflate_shutdown();
	return ret;
}
