/*
 * hd_png_host.c -- hd7png: the IDAT re-coder of applet/7png.c (:74-404) over libhipdeflate.so.
 *
 *     hd7png -G6 [-t] < before.png > after.png                 (the reference's filter form, applet/7png.c:421)
 *     hd7png -G6 [-t] in1.png out1.png in2.png out2.png ...    (many images, ONE inflate batch + ONE deflate batch)
 *
 * What the reference does per image: collect the IDAT chunks (:97-110), inflate their concatenation with zlib
 * (:116-292), code the pixels again through zlibutil_buffer_code with rfc1950 = 1 (:296-331) and write one IDAT +
 * IEND (:361-365); with -t only IHDR, PLTE, tRNS, IDAT and IEND are kept (:368).  Here the same two steps are two
 * batch calls over every image of the command line: hipdeflate_batch_inflate on the raw DEFLATE inside the zlib
 * streams (the size of the pixels comes from IHDR; the Adler-32 is checked on the host), then
 * hipdeflate_batch_deflate in HD_FRAME_ZLIB (78 da, raw DEFLATE -- long images in flushed segments, one wavefront
 * each --, Adler-32 from the device).  stderr keeps the reference's lines ("IDAT data length=", "compressed
 * length=", "recompressed length=", "Done.").  Not here: Apple's CgBI variant (:119,:280-292), which needs the
 * BGRA swap of the pixels; such a file is refused.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hipdeflate.h"
#include "zlibutil_hip.h"

static uint32_t be32(const unsigned char *p) { return (uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]; }
static void put_be32(unsigned char *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

struct image {
	const char *in_path, *out_path;
	unsigned char *file;            /* the whole input */
	size_t file_len;
	unsigned char *idat;            /* concatenated IDAT payloads = one zlib stream */
	size_t idat_len;
	size_t first_idat, after_idat;  /* chunk offsets in `file`: where the IDATs start / the first chunk behind them */
	uint64_t raw_len;               /* bytes of filtered scanlines, from IHDR */
};

/* bytes of the filtered scanlines: H x (1 + ceil(W x bits / 8)), per Adam7 pass when interlaced (PNG 1.2, 8.2) */
static uint64_t png_raw_size(uint32_t w, uint32_t h, unsigned depth, unsigned color, unsigned interlace)
{
	static const unsigned chan[7] = { 1, 0, 3, 1, 2, 0, 4 };
	if (color > 6 || !chan[color] || !w || !h)
		return 0;
	const uint64_t bits = (uint64_t)depth * chan[color];
	if (!interlace)
		return (uint64_t)h * (1 + (w * bits + 7) / 8);
	static const unsigned x0[7] = { 0, 4, 0, 2, 0, 1, 0 }, y0[7] = { 0, 0, 4, 0, 2, 0, 1 };
	static const unsigned dx[7] = { 8, 8, 4, 4, 2, 2, 1 }, dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
	uint64_t n = 0;
	for (int p = 0; p < 7; p++) {
		const uint64_t pw = w > x0[p] ? (w - x0[p] + dx[p] - 1) / dx[p] : 0, ph = h > y0[p] ? (h - y0[p] + dy[p] - 1) / dy[p] : 0;
		if (pw && ph)
			n += ph * (1 + (pw * bits + 7) / 8);
	}
	return n;
}

static int load(struct image *im, FILE *f)
{
	size_t cap = 1 << 20;
	im->file = (unsigned char *)malloc(cap);
	for (;;) {
		const size_t r = fread(im->file + im->file_len, 1, cap - im->file_len, f);
		im->file_len += r;
		if (im->file_len < cap)
			break;
		im->file = (unsigned char *)realloc(im->file, cap *= 2);
	}
	if (im->file_len < 8 || memcmp(im->file, "\x89PNG\x0d\x0a\x1a\x0a", 8)) {
		fprintf(stderr, "not PNG file\n");                       /* applet/7png.c:79-82 */
		return -1;
	}
	im->idat = (unsigned char *)malloc(im->file_len);
	size_t pos = 8;
	int seen_ihdr = 0;
	while (pos + 12 <= im->file_len) {
		const uint32_t len = be32(im->file + pos);
		const unsigned char *type = im->file + pos + 4;
		if ((size_t)len + 12 > im->file_len - pos)
			break;
		if (!memcmp(type, "CgBI", 4)) {
			fprintf(stderr, "Sorry: CgBI (Apple) images need the BGRA conversion of applet/7png.c:280-292; not handled here.\n");
			return -1;
		}
		if (!memcmp(type, "IHDR", 4) && len >= 13) {
			const unsigned char *d = im->file + pos + 8;
			im->raw_len = png_raw_size(be32(d), be32(d + 4), d[8], d[9], d[12]);
			seen_ihdr = 1;
		}
		if (!memcmp(type, "IDAT", 4)) {
			fprintf(stderr, "IDAT data length=%u\n", len);       /* :103 */
			if (!im->first_idat)
				im->first_idat = pos;
			memcpy(im->idat + im->idat_len, im->file + pos + 8, len);
			im->idat_len += len;
			im->after_idat = pos + 12 + len;
		}
		if (!memcmp(type, "IEND", 4))
			break;
		pos += 12 + (size_t)len;
	}
	if (!seen_ihdr || !im->raw_len || im->raw_len > 0xfff00000ull || im->idat_len < 6) {
		fprintf(stderr, "PNG without a usable IHDR / IDAT\n");
		return -1;
	}
	/* the dimensions come from the file: DEFLATE expands at most 1032 : 1 (258 bytes per 2 bits), so an IHDR that asks
	 * for more than its IDAT can hold is refused before anything of that size is allocated */
	if (im->raw_len > 1032ull * im->idat_len + 64) {
		fprintf(stderr, "IDAT is too short for the image the IHDR describes\n");
		return -1;
	}
	if ((im->idat[0] & 0x0f) != 8 || ((im->idat[0] << 8 | im->idat[1]) % 31) || (im->idat[1] & 0x20)) {
		fprintf(stderr, "IDAT is not a zlib stream\n");
		return -1;
	}
	fprintf(stderr, "compressed length=%d\n", (int)im->idat_len);      /* :111 */
	return 0;
}

static void put_chunk(FILE *f, const char *type, const unsigned char *data, uint32_t len)
{
	unsigned char b[4];
	put_be32(b, len);
	fwrite(b, 1, 4, f);
	fwrite(type, 1, 4, f);
	if (len)
		fwrite(data, 1, len, f);
	unsigned int crc = hd_crc32(0, (const unsigned char *)type, 4);     /* applet/7png.c:58-61 */
	if (len)
		crc = hd_crc32(crc, data, len);
	put_be32(b, crc);
	fwrite(b, 1, 4, f);
}

static int keep_when_stripping(const unsigned char *type)
{
	return !memcmp(type, "IHDR", 4) || !memcmp(type, "PLTE", 4) || !memcmp(type, "tRNS", 4);     /* :368 */
}

static void write_png(const struct image *im, FILE *f, const unsigned char *z, uint32_t zlen, int strip)
{
	fwrite(im->file, 1, 8, f);
	size_t pos = 8;
	int idat_done = 0;
	while (pos + 12 <= im->file_len) {
		const uint32_t len = be32(im->file + pos);
		const unsigned char *type = im->file + pos + 4;
		if ((size_t)len + 12 > im->file_len - pos)
			break;
		if (!memcmp(type, "IDAT", 4)) {
			if (!idat_done)
				put_chunk(f, "IDAT", z, zlen);                 /* :361 */
			idat_done = 1;
		} else if (!memcmp(type, "IEND", 4)) {
			put_chunk(f, "IEND", NULL, 0);                         /* :365 */
			break;
		} else if (!strip || keep_when_stripping(type)) {
			fwrite(im->file + pos, 1, 12 + (size_t)len, f);         /* copied with its CRC */
		}
		pos += 12 + (size_t)len;
	}
}

int main(int argc, char **argv)
{
	int level = -1, strip = 0, nfiles = 0;
	const char *paths[512];
	for (int i = 1; i < argc; i++) {
		const char *a = argv[i];
		if (!strncmp(a, "-G", 2) || !strncmp(a, "-l", 2))
			level = a[2] ? atoi(a + 2) : 6;
		else if (!strcmp(a, "-t") || !strcmp(a, "--strip"))
			strip = 1;
		else if (!strcmp(a, "-c"))
			;
		else if (a[0] != '-' && nfiles < 512)
			paths[nfiles++] = a;
		else
			level = -2;
	}
	if (level < 0 || (nfiles & 1)) {
		fprintf(stderr, "usage: %s -G<level> [-t] < before.png > after.png   or   -G<level> [-t] in.png out.png [in2.png out2.png ...]\n", argv[0]);
		return 1;
	}
	const uint32_t n = nfiles ? (uint32_t)nfiles / 2 : 1;
	struct image *im = (struct image *)calloc(n, sizeof(*im));
	for (uint32_t k = 0; k < n; k++) {
		FILE *f = nfiles ? fopen(paths[2 * k], "rb") : stdin;
		if (!f) {
			fprintf(stderr, "cannot open %s\n", paths[2 * k]);
			return 1;
		}
		im[k].in_path = nfiles ? paths[2 * k] : "-";
		im[k].out_path = nfiles ? paths[2 * k + 1] : "-";
		if (load(&im[k], f))
			return 1;
		if (nfiles)
			fclose(f);
	}
	if (hipdeflate_init(-1))
		return 1;
	fprintf(stderr, "compression level = %d (hip)\n", level);

	/* ---- every image's pixels in one inflate batch ------------------------------------------------------------- */
	uint64_t *zoff = (uint64_t *)malloc(n * 8), *roff = (uint64_t *)malloc(n * 8);
	uint32_t *zlen = (uint32_t *)malloc(n * 4), *rcap = (uint32_t *)malloc(n * 4), *rlen = (uint32_t *)malloc(n * 4);
	int32_t *st = (int32_t *)malloc(n * 4);
	size_t ztotal = 0, rtotal = 0;
	for (uint32_t k = 0; k < n; k++) {
		zoff[k] = ztotal;
		zlen[k] = (uint32_t)(im[k].idat_len - 2);                  /* raw DEFLATE + the Adler-32 as "trailing bytes" */
		ztotal += (im[k].idat_len + 15) & ~(size_t)15;
		roff[k] = rtotal;
		rcap[k] = (uint32_t)im[k].raw_len;
		rtotal += (im[k].raw_len + 15) & ~(size_t)15;
	}
	unsigned char *zin = (unsigned char *)calloc(1, ztotal + 16), *raw = (unsigned char *)malloc(rtotal + 16);
	for (uint32_t k = 0; k < n; k++)
		memcpy(zin + zoff[k], im[k].idat + 2, im[k].idat_len - 2);
	if (hipdeflate_batch_inflate(zin, zoff, zlen, n, raw, roff, rcap, rlen, NULL, st)) {
		fprintf(stderr, "inflate: the batch did not run\n");
		return 1;
	}
	for (uint32_t k = 0; k < n; k++) {
		const unsigned char *t = im[k].idat + im[k].idat_len - 4;
		if (st[k] || rlen[k] != rcap[k] || hd_adler32(1, raw + roff[k], rlen[k]) != be32(t)) {
			fprintf(stderr, "%s: inflate %d (pixels %u of %u bytes, Adler-32 %s)\n", im[k].in_path, st[k], rlen[k], rcap[k],
				st[k] || rlen[k] != rcap[k] ? "not checked" : "differs");
			return 1;
		}
	}
	/* ---- ... and one deflate batch, RFC 1950 members straight from the device ---------------------------------- */
	uint32_t maxraw = 0;
	for (uint32_t k = 0; k < n; k++)
		maxraw = rcap[k] > maxraw ? rcap[k] : maxraw;
	const uint64_t slot = hipdeflate_bound(maxraw, level);
	unsigned char *out = (unsigned char *)malloc((size_t)slot * n);
	uint32_t *olen = (uint32_t *)malloc(n * 4);
	if (!out || hipdeflate_batch_deflate(raw, roff, rcap, n, level, HD_FRAME_ZLIB, out, slot, slot > 0xfffffff0u ? 0xfffffff0u : (uint32_t)slot,
					     olen, NULL, st)) {
		fprintf(stderr, "hip_deflate: the batch did not run\n");
		return 1;
	}
	for (uint32_t k = 0; k < n; k++) {
		if (st[k]) {
			fprintf(stderr, "hip_deflate %d\n", st[k]);
			return 1;
		}
		FILE *f = nfiles ? fopen(im[k].out_path, "wb") : stdout;
		if (!f) {
			fprintf(stderr, "cannot open %s\n", im[k].out_path);
			return 1;
		}
		write_png(&im[k], f, out + (size_t)k * slot, olen[k], strip);
		fprintf(stderr, "recompressed length=%d\n", (int)olen[k]);       /* :362 */
		if (nfiles)
			fclose(f);
	}
	fflush(stdout);
	fprintf(stderr, "Done.\n");
	hipdeflate_shutdown();
	return 0;
}
