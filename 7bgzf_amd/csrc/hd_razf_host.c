/*
 * hd_razf_host.c -- hd7razf: applet/7razf.c (_compress :160-304, _decompress
 * :306-383) over libhipdeflate.so.  RAZF is the random-access gzip of old samtools:
 * ONE gzip member with an 'RAZF' extra field, 32 KiB chunks of which every one but
 * the last ends in a full flush (the last is an ordinary final stream), CRC-32 and
 * ISIZE, then a big-endian index (chunk count - 1, 64-bit bin offsets every 2^32
 * input bytes, 32-bit chunk offsets inside the bin) and two 64-bit totals.
 *
 *     hd7razf -G<level> dec.bin > enc.raz
 *     hd7razf -d enc.raz > dec.bin
 *
 * What changed, and why: as hd7dictzip -- chunks go to the device in batches and
 * leave in full-flush form (HD_FRAME_RAW_FLUSH; the last chunk HD_FRAME_RAW), no
 * re-inflate on the CPU (zlibutil_buffer_full_flush, applet/7razf.c:126-160), the
 * member CRC-32 is folded from the kernel's per-chunk CRCs, and the reader checks
 * CRC-32 and size, which the reference's does not.  The first chunk has index -1
 * and no cell in the table, as there (:177,:277).
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hipdeflate.h"
#include "hd_host_util.h"

#define RZ_BLOCK 32768u
#define RZ_BATCH 4096                 /* chunks per device call */
#define RZ_BINSIZE ((1ull << 32) / RZ_BLOCK)

static const unsigned char rz_header[19] = { 0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0x00, 0x03, 0x07, 0x00,
					     'R',  'A',  'Z',  'F',  0x01, 0x80, 0x00 };

static int rz_compress(FILE *in, FILE *out, int level)
{
	const long long total = file_size(in);
	if (total <= 0) {
		fprintf(stderr, total ? "cannot stat the input\n" : "empty input\n");
		return 2;
	}
	const long long nchunks = (total + RZ_BLOCK - 1) / RZ_BLOCK;
	const long long total_block = nchunks - 1;          /* the reference's count: chunk -1 is not indexed */
	const long long bins = total_block / (long long)RZ_BINSIZE;
	const size_t stride = up16(RZ_BLOCK + 5 * 2 + 32);
	const size_t index_bytes = 4 + 8 * (size_t)(bins + 1) + 4 * (size_t)total_block;
	unsigned char *ibuf = malloc((size_t)RZ_BATCH * RZ_BLOCK + 16);
	unsigned char *obuf = malloc((size_t)RZ_BATCH * stride + 16);
	uint64_t *off = malloc(sizeof(uint64_t) * RZ_BATCH);
	uint32_t *len = malloc(sizeof(uint32_t) * RZ_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * RZ_BATCH);
	uint32_t *crc = malloc(sizeof(uint32_t) * RZ_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * RZ_BATCH);
	unsigned char *index = calloc(1, index_bytes + 16);
	if (!ibuf || !obuf || !off || !len || !olen || !crc || !st || !index) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	unsigned char *bin_tab = index + 4, *cell_tab = index + 4 + 8 * (size_t)(bins + 1);
	wr32be(index, (uint32_t)total_block);
	fwrite(rz_header, 1, sizeof(rz_header), out);
	uint64_t pos = sizeof(rz_header);
	struct crc_fold fold = { 0, 0, 0 };
	long long left = total;
	int ret = 0;
	for (long long c = 0; c < nchunks && !ret; c += RZ_BATCH) {
		uint32_t n = (uint32_t)(nchunks - c < RZ_BATCH ? nchunks - c : RZ_BATCH);
		const int has_last = c + n == nchunks;
		const size_t want = left < (long long)n * RZ_BLOCK ? (size_t)left : (size_t)n * RZ_BLOCK;
		if (fread(ibuf, 1, want, in) != want) {
			fprintf(stderr, "short read\n");
			ret = 2;
			break;
		}
		for (uint32_t i = 0; i < n; i++) {
			off[i] = (uint64_t)i * RZ_BLOCK;
			const size_t rest = want - (size_t)off[i];
			len[i] = rest < RZ_BLOCK ? (uint32_t)rest : RZ_BLOCK;
		}
		/* every chunk but the file's last in full-flush form; the last one is a finished stream (:226-236) */
		const uint32_t nflush = has_last ? n - 1 : n;
		int r = 0;
		if (nflush)
			r = hipdeflate_batch_deflate(ibuf, off, len, nflush, level, HD_FRAME_RAW_FLUSH, obuf, stride, (uint32_t)stride,
						     olen, crc, st);
		if (!r && has_last)
			r = hipdeflate_batch_deflate(ibuf, off + nflush, len + nflush, 1, level, HD_FRAME_RAW,
						     obuf + (size_t)nflush * stride, stride, (uint32_t)stride, olen + nflush,
						     crc + nflush, st + nflush);
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
			const long long idx = c + i - 1;      /* the reference's i: -1 for the first chunk */
			if (idx >= 0) {
				if (idx % (long long)RZ_BINSIZE == 0)
					wr64be(bin_tab + 8 * (size_t)(idx / (long long)RZ_BINSIZE), pos);
				wr32be(cell_tab + 4 * (size_t)idx, (uint32_t)(pos - rd64be(bin_tab + 8 * (size_t)(idx / (long long)RZ_BINSIZE))));
			}
			fwrite(obuf + (size_t)i * stride, 1, olen[i], out);
			pos += olen[i];
			crc_append(&fold, crc[i], len[i]);
		}
		left -= (long long)want;
		fprintf(stderr, "%lld / %lld\r", c + n - 1, total_block);
	}
	if (!ret) {
		unsigned char t[16];
		wr32(t, fold.crc);
		wr32(t + 4, (uint32_t)total);
		fwrite(t, 1, 8, out);
		pos += 8;
		fwrite(index, 1, index_bytes, out);
		wr64be(t, (uint64_t)total);
		wr64be(t + 8, pos);                         /* where the index starts */
		fwrite(t, 1, 16, out);
		fprintf(stderr, "%lld / %lld done.\n", total_block, total_block);
		if (fflush(out) || ferror(out)) {
			fprintf(stderr, "write error\n");
			ret = 2;
		}
	}
	free(ibuf), free(obuf), free(off), free(len), free(olen), free(crc), free(st), free(index);
	return ret;
}

static int rz_decompress(FILE *in, FILE *out)
{
	unsigned char head[64], tail[16];
	const long long fsize = file_size(in);
	if (fread(head, 1, 19, in) != 19 || memcmp(head, rz_header, 4) || !(head[3] & 4) || rd16(head + 10) < 7 ||
	    memcmp(head + 12, "RAZF", 4)) {
		fprintf(stderr, "not RAZF\n");
		return 1;
	}
	const uint32_t block_size = (head[17] << 8) | head[18];
	const size_t hdr_len = 12 + rd16(head + 10);
	if (!block_size || fsize < (long long)hdr_len + 8 + 4 + 8 + 16 || fseeko(in, -16, SEEK_END)) {
		fprintf(stderr, "input is not seekable or truncated\n");
		return 1;
	}
	if (fread(tail, 1, 16, in) != 16)
		return 1;
	const uint64_t total_bytes = rd64be(tail), index_at = rd64be(tail + 8);
	if (index_at + 4 + 8 + 16 > (uint64_t)fsize || index_at < hdr_len + 8) {
		fprintf(stderr, "corrupted index\n");
		return 1;
	}
	fseeko(in, (long long)index_at, SEEK_SET);
	if (fread(head + 32, 1, 4, in) != 4)
		return 1;
	const long long total_block = (int32_t)rd32be(head + 32);
	const uint64_t binsize = (1ull << 32) / block_size;
	const long long bins = total_block / (long long)binsize;
	const size_t tab_bytes = 8 * (size_t)(bins + 1) + 4 * (size_t)total_block;
	if (total_block < 0 || index_at + 4 + tab_bytes + 16 != (uint64_t)fsize) {
		fprintf(stderr, "corrupted index\n");
		return 1;
	}
	unsigned char *tab = malloc(tab_bytes + 16);
	if (!tab || fread(tab, 1, tab_bytes, in) != tab_bytes) {
		fprintf(stderr, "corrupted index\n");
		return 1;
	}
	const unsigned char *cell_tab = tab + 8 * (size_t)(bins + 1);
	const long long nchunks = total_block + 1;
	/* start of chunk k (k = the reference's i + 1); chunk nchunks "starts" at the trailer's end (:325) */
#define RZ_START(k) ((k) == 0 ? (uint64_t)hdr_len : (k) == nchunks ? index_at : \
		     rd64be(tab + 8 * (size_t)(((k) - 1) / (long long)binsize)) + rd32be(cell_tab + 4 * (size_t)((k) - 1)))
	unsigned char *ibuf = NULL, *obuf = malloc((size_t)RZ_BATCH * up16(block_size) + 16);
	size_t icap = 0;
	uint64_t *ioff = malloc(sizeof(uint64_t) * RZ_BATCH), *ooff = malloc(sizeof(uint64_t) * RZ_BATCH);
	uint32_t *ilen = malloc(sizeof(uint32_t) * RZ_BATCH), *cap = malloc(sizeof(uint32_t) * RZ_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * RZ_BATCH), *crc = malloc(sizeof(uint32_t) * RZ_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * RZ_BATCH);
	if (!obuf || !ioff || !ooff || !ilen || !cap || !olen || !crc || !st) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	struct crc_fold fold = { 0, 0, 0 };
	uint64_t produced = 0;
	int ret = 0;
	for (long long c = 0; c < nchunks && !ret; c += RZ_BATCH) {
		const uint32_t m = (uint32_t)(nchunks - c < RZ_BATCH ? nchunks - c : RZ_BATCH);
		const uint64_t first = RZ_START(c), end = RZ_START(c + m);
		if (end < first || end > index_at) {
			fprintf(stderr, "corrupted index\n");
			ret = 1;
			break;
		}
		const size_t itotal = (size_t)(end - first);
		if (itotal + 16 > icap) {
			free(ibuf);
			ibuf = malloc(icap = itotal + 16);
			if (!ibuf) {
				fprintf(stderr, "out of memory\n");
				ret = 2;
				break;
			}
		}
		fseeko(in, (long long)first, SEEK_SET);
		if (fread(ibuf, 1, itotal, in) != itotal) {
			fprintf(stderr, "unexpected end of file\n");
			ret = 1;
			break;
		}
		for (uint32_t i = 0; i < m && !ret; i++) {
			const uint64_t a = RZ_START(c + i), b = RZ_START(c + i + 1);
			if (a < first || b < a || b > end)
				ret = 1;
			ioff[i] = a - first;
			ilen[i] = (uint32_t)(b - a);          /* the last chunk's span takes the 8 trailer bytes along, as there */
			ooff[i] = (uint64_t)i * up16(block_size);
			cap[i] = block_size;
		}
		if (ret) {
			fprintf(stderr, "corrupted index\n");
			break;
		}
		int r = hipdeflate_batch_inflate_flush(ibuf, ioff, ilen, m, obuf, ooff, cap, olen, crc, st);
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
			fwrite(obuf + ooff[i], 1, olen[i], out);
			crc_append(&fold, crc[i], olen[i]);
			produced += olen[i];
		}
		fprintf(stderr, "%lld / %lld\r", c + m - 1, total_block);
	}
	if (!ret) {
		fprintf(stderr, "%lld / %lld done.\n", total_block, total_block);
		unsigned char t[8];
		fseeko(in, (long long)index_at - 8, SEEK_SET);
		if (fread(t, 1, 8, in) != 8 || rd32(t) != fold.crc || rd32(t + 4) != (uint32_t)produced || produced != total_bytes) {
			fprintf(stderr, "crc32 / size mismatch\n");
			ret = 1;
		}
	}
	if (!ret && (fflush(out) || ferror(out))) {
		fprintf(stderr, "write error\n");
		ret = 2;
	}
	free(tab), free(ibuf), free(obuf), free(ioff), free(ooff), free(ilen), free(cap), free(olen), free(crc), free(st);
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
	if (bad || !name || (decode && level >= 0) || (!decode && (level < 0 || level > 9))) {
		fprintf(stderr, "usage: %s -G<level> dec.bin > enc.raz   or   -d enc.raz > dec.bin\n", argv[0]);
		return 1;
	}
	int r = hipdeflate_init(-1);
	if (r) {
		fprintf(stderr, "hipdeflate: no usable device (%d): %s\n", r, hipdeflate_version());
		return 4;
	}
	const double t0 = now_s();
	FILE *in = fopen(name, "rb");
	if (!in) {
		fprintf(stderr, "failed to open %s\n", name);
		return 2;
	}
	int ret;
	if (decode) {
		ret = rz_decompress(in, stdout);
	} else {
		fprintf(stderr, "compression level = %d (hip)\n", level);
		ret = rz_compress(in, stdout, level);
	}
	fclose(in);
	fprintf(stderr, "ellapsed time: %.3f sec\n", now_s() - t0);
	hipdeflate_shutdown();
	return ret;
}
