/*
 * hd_dictzip_host.c -- hd7dictzip: applet/7dictzip.c (_compress :179-325,
 * _decompress :327-402) over libhipdeflate.so.  Same command line shape and the
 * same file format ("improved dictzip": one gzip member per <= 32762 chunks, an
 * 'RA' extra field holding the compressed size of every chunk, chunks in
 * full-flush form, then an empty final block `03 00`, CRC-32 and ISIZE):
 *
 *     hd7dictzip -G<level> [-X] dec.bin enc.dz       (-X: 0xff00-byte chunks, else 58315)
 *     hd7dictzip -d enc.dz > dec.bin
 *
 * What changed, and why: the reference compresses one chunk per pthread with a
 * final block and then re-inflates it on the CPU with a patched zlib to turn it
 * into full-flush form (zlibutil_buffer_full_flush, :93-126).  Here a batch of
 * chunks goes to the device in one call and the kernel emits the full-flush form
 * itself (HD_FRAME_RAW_FLUSH); the CRC-32 of the whole member is folded from the
 * per-chunk CRCs the kernel returns (the reference runs fcrc32 over the input on
 * the host, :203), and the reader checks it, which the reference's does not.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hipdeflate.h"
#include "hd_host_util.h"

#define DZ_BATCH 2048                 /* chunks per device call */
#define DZ_MAX_CHUNKS 32762           /* (0xffff - 10) / 2, applet/7dictzip.c:180 */

/* ---- compress ---------------------------------------------------------------------- */

static int dz_compress(FILE *in, FILE *out, int level, uint32_t block_size)
{
	const long long max_member = (long long)block_size * DZ_MAX_CHUNKS;
	const long long total = file_size(in);
	if (total < 0) {
		fprintf(stderr, "cannot stat the input\n");
		return 2;
	}
	const size_t stride = up16((size_t)block_size + 5 * (block_size / 65535 + 1) + 32);
	unsigned char *ibuf = malloc((size_t)DZ_BATCH * block_size + 16);
	unsigned char *obuf = malloc((size_t)DZ_BATCH * stride + 16);
	uint64_t *off = malloc(sizeof(uint64_t) * DZ_BATCH);
	uint32_t *len = malloc(sizeof(uint32_t) * DZ_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * DZ_BATCH);
	uint32_t *crc = malloc(sizeof(uint32_t) * DZ_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * DZ_BATCH);
	unsigned char *sizes = malloc(2 * DZ_MAX_CHUNKS + 16);
	if (!ibuf || !obuf || !off || !len || !olen || !crc || !st || !sizes) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	long long done = 0;
	int ret = 0;
	do {
		const long long cur = total - done < max_member ? total - done : max_member;
		const uint32_t nchunks = (uint32_t)((cur + block_size - 1) / block_size);
		unsigned char hdr[22] = { 0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0x00, 0x03 };
		wr16(hdr + 10, 10 + 2 * nchunks);
		hdr[12] = 'R', hdr[13] = 'A';
		wr16(hdr + 14, 6 + 2 * nchunks);
		wr16(hdr + 16, 1);
		wr16(hdr + 18, block_size);
		wr16(hdr + 20, nchunks);
		fwrite(hdr, 1, 22, out);
		const long long pos_sizes = ftello(out);
		memset(sizes, 0, 2 * (size_t)nchunks);
		fwrite(sizes, 1, 2 * (size_t)nchunks, out);

		struct crc_fold fold = { 0, 0, 0 };
		long long left = cur;
		for (uint32_t c = 0; c < nchunks && !ret; c += DZ_BATCH) {
			const uint32_t n = nchunks - c < DZ_BATCH ? nchunks - c : DZ_BATCH;
			const size_t want = left < (long long)n * block_size ? (size_t)left : (size_t)n * block_size;
			if (fread(ibuf, 1, want, in) != want) {
				fprintf(stderr, "short read\n");
				ret = 2;
				break;
			}
			for (uint32_t i = 0; i < n; i++) {
				off[i] = (uint64_t)i * block_size;
				const size_t rest = want - (size_t)off[i];
				len[i] = rest < block_size ? (uint32_t)rest : block_size;
			}
			int r = hipdeflate_batch_deflate(ibuf, off, len, n, level, HD_FRAME_RAW_FLUSH, obuf, stride, 65535, olen, crc,
							 st);
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
				wr16(sizes + 2 * (size_t)(c + i), olen[i]);
				fwrite(obuf + (size_t)i * stride, 1, olen[i], out);
				crc_append(&fold, crc[i], len[i]);
			}
			left -= (long long)want;
			fprintf(stderr, "%u / %u\r", c + n, nchunks);
		}
		if (ret)
			break;
		const long long pos = ftello(out);
		fseeko(out, pos_sizes, SEEK_SET);
		fwrite(sizes, 1, 2 * (size_t)nchunks, out);
		fseeko(out, pos, SEEK_SET);
		unsigned char trl[10] = { 0x03, 0x00 };     /* empty final block: a plain gzip reader sees a whole member */
		wr32(trl + 2, fold.crc);
		wr32(trl + 6, (uint32_t)cur);
		fwrite(trl, 1, 10, out);
		fprintf(stderr, "%u / %u done.\n", nchunks, nchunks);
		done += cur;
	} while (done < total);
	if (!ret && (fflush(out) || ferror(out))) {
		fprintf(stderr, "write error\n");
		ret = 2;
	}
	free(ibuf), free(obuf), free(off), free(len), free(olen), free(crc), free(st), free(sizes);
	return ret;
}

/* ---- decompress -------------------------------------------------------------------- */

/* applet/7dictzip.c:137-177: the gzip header with the 'RA' field; returns the header
 * length, 0 if this is not such a header */
static size_t dz_header(const unsigned char *d, size_t size, size_t *sizes_off, uint32_t *block_size, uint32_t *nchunks)
{
	if (size < 12 || d[0] != 0x1f || d[1] != 0x8b || d[2] != 8 || (d[3] & 0xe0) || !(d[3] & 4))
		return 0;
	const uint32_t flags = d[3];
	size_t n = 10;
	const uint32_t xlen = rd16(d + n);
	n += 2;
	if (size < n + xlen || xlen < 10)
		return 0;
	if (!(d[n] == 'R' && d[n + 1] == 'A' && rd16(d + n + 2) + 4 == xlen && d[n + 4] == 1 && d[n + 5] == 0 &&
	      rd16(d + n + 8) * 2 + 10 == xlen))
		return 0;
	*block_size = rd16(d + n + 6);
	*nchunks = rd16(d + n + 8);
	*sizes_off = n + 10;
	n += xlen;
	if (flags & 8)
		while (n < size && d[n++])
			;
	if (flags & 16)
		while (n < size && d[n++])
			;
	if (flags & 2)
		n += 2;
	return n <= size ? n : 0;
}

static int dz_decompress(FILE *in, FILE *out)
{
	unsigned char *head = malloc(65536 + 280);
	unsigned char *ibuf = NULL, *obuf = NULL;
	size_t icap = 0, ocap = 0;
	uint64_t *ioff = malloc(sizeof(uint64_t) * DZ_BATCH), *ooff = malloc(sizeof(uint64_t) * DZ_BATCH);
	uint32_t *ilen = malloc(sizeof(uint32_t) * DZ_BATCH), *cap = malloc(sizeof(uint32_t) * DZ_BATCH);
	uint32_t *olen = malloc(sizeof(uint32_t) * DZ_BATCH), *crc = malloc(sizeof(uint32_t) * DZ_BATCH);
	int32_t *st = malloc(sizeof(int32_t) * DZ_BATCH);
	const long long fsize = file_size(in);
	int ret = 0;
	if (!head || !ioff || !ooff || !ilen || !cap || !olen || !crc || !st) {
		fprintf(stderr, "out of memory\n");
		return 2;
	}
	while (!ret) {
		const long long pos = ftello(in);
		if (pos >= fsize)
			break;
		const size_t got = fread(head, 1, 65536 + 280, in);
		size_t sizes_off = 0;
		uint32_t block_size = 0, nchunks = 0;
		const size_t n = dz_header(head, got, &sizes_off, &block_size, &nchunks);
		if (!n || !block_size) {
			fprintf(stderr, "header is not gzip (possibly corrupted)\n");
			ret = 1;
			break;
		}
		const unsigned char *sizes = head + sizes_off;
		fseeko(in, pos + (long long)n, SEEK_SET);
		struct crc_fold fold = { 0, 0, 0 };
		uint64_t produced = 0;
		for (uint32_t c = 0; c < nchunks && !ret; c += DZ_BATCH) {
			const uint32_t m = nchunks - c < DZ_BATCH ? nchunks - c : DZ_BATCH;
			size_t itotal = 0;
			for (uint32_t i = 0; i < m; i++) {
				ioff[i] = itotal;
				ilen[i] = rd16(sizes + 2 * (size_t)(c + i));
				itotal += ilen[i];
				ooff[i] = (uint64_t)i * up16(block_size);
				cap[i] = block_size;
			}
			if (itotal + 16 > icap) {
				free(ibuf);
				ibuf = malloc(icap = itotal + 16);
			}
			if ((size_t)m * up16(block_size) + 16 > ocap) {
				free(obuf);
				obuf = malloc(ocap = (size_t)m * up16(block_size) + 16);
			}
			if (!ibuf || !obuf) {
				fprintf(stderr, "out of memory\n");
				ret = 2;
				break;
			}
			if (fread(ibuf, 1, itotal, in) != itotal) {
				fprintf(stderr, "unexpected end of file\n");
				ret = 1;
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
			fprintf(stderr, "%u / %u\r", c + m, nchunks);
		}
		if (ret)
			break;
		fprintf(stderr, "%u / %u done.\n", nchunks, nchunks);
		/* trailer: `03 00` (ours and the reference's writer) or nothing (classic dictzip, whose last
		 * chunk carries the final block), then CRC-32 and ISIZE; applet/7dictzip.c:393-399 */
		unsigned char t[12];
		const size_t tn = fread(t, 1, 12, in);
		size_t at = 0;
		if (tn >= 10 && t[0] == 0x03 && t[1] == 0x00 && (tn == 10 || (tn == 12 && t[10] == 0x1f && t[11] == 0x8b)))
			at = 2;
		if (tn < at + 8) {
			fprintf(stderr, "unexpected end of file\n");
			ret = 1;
			break;
		}
		if (rd32(t + at) != fold.crc || rd32(t + at + 4) != (uint32_t)produced) {
			fprintf(stderr, "crc32 / size mismatch\n");
			ret = 1;
			break;
		}
		fseeko(in, -(long long)(tn - at - 8), SEEK_CUR);
	}
	if (!ret && (fflush(out) || ferror(out))) {
		fprintf(stderr, "write error\n");
		ret = 2;
	}
	free(head), free(ibuf), free(obuf), free(ioff), free(ooff), free(ilen), free(cap), free(olen), free(crc), free(st);
	return ret;
}

int main(int argc, char **argv)
{
	int level = -1, decode = 0, extreme = 0;
	const char *names[2] = { NULL, NULL };
	int nn = 0;
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
				else if (*p == 'X')
					extreme = 1;
				else if (*p == 'G' || *p == 'l') {
					level = p[1] ? atoi(p + 1) : 1;
					break;
				} else {
					nn = 3;
					break;
				}
			}
		} else if (nn < 2) {
			names[nn++] = a;
		} else {
			nn = 3;
		}
	}
	if (nn == 3 || (decode && (nn != 1 || level >= 0)) || (!decode && (nn != 2 || level < 0 || level > 9))) {
		fprintf(stderr, "usage: %s -G<level> [-X] dec.bin enc.dz   or   -d enc.dz > dec.bin\n", argv[0]);
		return 1;
	}
	int r = hipdeflate_init(-1);
	if (r) {
		fprintf(stderr, "hipdeflate: no usable device (%d): %s\n", r, hipdeflate_version());
		return 4;
	}
	const double t0 = now_s();
	int ret;
	FILE *in = fopen(names[0], "rb");
	if (!in) {
		fprintf(stderr, "failed to open %s\n", names[0]);
		return 2;
	}
	if (decode) {
		ret = dz_decompress(in, stdout);
	} else {
		FILE *out = fopen(names[1], "wb");
		if (!out) {
			fprintf(stderr, "failed to open %s\n", names[1]);
			fclose(in);
			return 3;
		}
		fprintf(stderr, "compression level = %d (hip)\n", level);
		ret = dz_compress(in, out, level, extreme ? 0xff00 : 58315);
		if (fclose(out) && !ret)
			ret = 2;
	}
	fclose(in);
	fprintf(stderr, "ellapsed time: %.3f sec\n", now_s() - t0);
	hipdeflate_shutdown();
	return ret;
}
