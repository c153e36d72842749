/*
 * hd_checksum_frame.c -- oracle: checksums, stored encoder, container framing.
 * TEST INFRASTRUCTURE ONLY (see hd_oracle.h).  Plain C, bit-at-a-time where
 * that is the clearest statement of the definition; speed is irrelevant here.
 */
#include <string.h>
#include "hd_oracle.h"

/* CRC-32/IEEE 802.3, reflected polynomial 0xEDB88320, init/xorout 0xFFFFFFFF:
 * the function both crc32_gzip_refl (lib/isa-l/crc/crc_base.c:235) and zlib's
 * crc32 (lib/zlib/crc32.c:1015) compute.  Stated by definition, one bit per
 * step, so that it cannot share a table bug with the kernel under test. */
uint32_t hdo_crc32(uint32_t crc, const uint8_t *buf, size_t n)
{
	uint32_t c = ~crc;
	for (size_t i = 0; i < n; i++) {
		c ^= buf[i];
		for (int k = 0; k < 8; k++)
			c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
	}
	return ~c;
}

/* Adler-32 by definition (RFC 1950 8.2; lib/zlib/adler32.c:128). */
uint32_t hdo_adler32(uint32_t adler, const uint8_t *buf, size_t n)
{
	uint32_t a = adler & 0xffff, b = adler >> 16;
	for (size_t i = 0; i < n; i++) {
		a = (a + buf[i]) % 65521u;
		b = (b + a) % 65521u;
	}
	return (b << 16) | a;
}

static void put16(uint8_t *p, uint32_t v) { p[0] = v & 0xff; p[1] = (v >> 8) & 0xff; }
static void put32(uint8_t *p, uint32_t v) { put16(p, v); put16(p + 2, v >> 16); }

/* lib/zlibutil.c:302-325: ceil(n/65535) stored blocks, BFINAL on the last;
 * n == 0 gives ZERO blocks and destLen 0 (the reference's loop body never
 * runs) -- kept, because the applets never call it with an empty block. */
int hdo_store_deflate(uint8_t *dest, size_t *destLen, const uint8_t *source,
		      size_t sourceLen)
{
	size_t blocks = (sourceLen + 65534) / 65535;
	if (*destLen < sourceLen + 5 * blocks)
		return -5; /* Z_BUF_ERROR */
	*destLen = 0;
	for (size_t i = 0; i < blocks; i++) {
		uint32_t blk = sourceLen < 65535 ? (uint32_t)sourceLen : 65535u;
		dest[0] = i + 1 < blocks ? 0x00 : 0x01;
		put16(dest + 1, blk);
		put16(dest + 3, ~blk);
		memcpy(dest + 5, source, blk);
		source += blk;
		sourceLen -= blk;
		dest += blk + 5;
		*destLen += blk + 5;
	}
	return 0;
}

/* applet/7bgzf.c:263-272 (and bgzf_compress.c:191-197): fixed 10-byte gzip
 * header with FLG.FEXTRA, XLEN=6, subfield 'B','C',SLEN=2, BSIZE=total-1. */
size_t hdo_bgzf_frame(uint8_t *dst, size_t cap, const uint8_t *payload,
		      size_t payload_len, uint32_t crc, uint32_t isize)
{
	static const uint8_t hdr[16] = { 0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0x00,
					 0xff, 0x06, 0x00, 'B', 'C', 0x02, 0x00 };
	size_t total = 18 + payload_len + 8;
	if (total > 65536 || total > cap)
		return 0;
	memcpy(dst, hdr, 16);
	put16(dst + 16, (uint32_t)(total - 1));
	memmove(dst + 18, payload, payload_len);
	put32(dst + 18 + payload_len, crc);
	put32(dst + 22 + payload_len, isize);
	return total;
}

size_t hdo_bgzf_eof(uint8_t *dst, size_t cap)
{
	static const uint8_t eof[28] = { 0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0x00, 0xff,
					 0x06, 0x00, 'B', 'C', 0x02, 0x00, 0x1b, 0x00,
					 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0 };
	if (cap < 28)
		return 0;
	memcpy(dst, eof, 28);
	return 28;
}

/* applet/7migz.c:224-233: XLEN=8, subfield 'M','Z',SLEN=4, u32 = payload size */
size_t hdo_migz_frame(uint8_t *dst, size_t cap, const uint8_t *payload,
		      size_t payload_len, uint32_t crc, uint32_t isize)
{
	static const uint8_t hdr[16] = { 0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0x00,
					 0xff, 0x08, 0x00, 'M', 'Z', 0x04, 0x00 };
	size_t total = 20 + payload_len + 8;
	if (total > cap)
		return 0;
	memcpy(dst, hdr, 16);
	put32(dst + 16, (uint32_t)payload_len);
	memmove(dst + 20, payload, payload_len);
	put32(dst + 20 + payload_len, crc);
	put32(dst + 24 + payload_len, isize);
	return total;
}

/* lib/zlibutil.c:374-397: 78 da, the codec's bytes, Adler-32 of the INPUT big-endian */
size_t hdo_zlib_frame(uint8_t *dst, size_t cap, const uint8_t *payload,
		      size_t payload_len, uint32_t adler)
{
	size_t total = 2 + payload_len + 4;
	if (total > cap)
		return 0;
	memmove(dst + 2, payload, payload_len);
	dst[0] = 0x78;
	dst[1] = 0xda;
	dst[2 + payload_len] = (uint8_t)(adler >> 24);
	dst[3 + payload_len] = (uint8_t)(adler >> 16);
	dst[4 + payload_len] = (uint8_t)(adler >> 8);
	dst[5 + payload_len] = (uint8_t)adler;
	return total;
}

/* lib/zlibutil.c:379-405: 1f 8b 08 00 <mtime> 02 00, the codec's bytes, CRC32, ISIZE */
size_t hdo_gzip_frame(uint8_t *dst, size_t cap, const uint8_t *payload,
		      size_t payload_len, uint32_t mtime, uint32_t crc, uint32_t isize)
{
	size_t total = 10 + payload_len + 8;
	if (total > cap)
		return 0;
	memmove(dst + 10, payload, payload_len);
	dst[0] = 0x1f; dst[1] = 0x8b; dst[2] = 0x08; dst[3] = 0x00;
	put32(dst + 4, mtime);
	dst[8] = 0x02; dst[9] = 0x00;
	put32(dst + 10 + payload_len, crc);
	put32(dst + 14 + payload_len, isize);
	return total;
}

static uint32_t get16(const uint8_t *p) { return p[0] | (p[1] << 8); }
static uint32_t get32(const uint8_t *p) { return get16(p) | (get16(p + 2) << 16); }

/* applet/7bgzf.c:81-131.  Walks FLG exactly as the reference does and maps the
 * five recognised extra fields to a member length. */
int hdo_read_gz_header(const uint8_t *data, int size, int *extra_off,
		       int *extra_len, long long *block_len)
{
	int n, flags;
	if (size < 4 || data[0] != 0x1f || data[1] != 0x8b)
		return 0;
	flags = data[3];
	if (data[2] != 8 || (flags & 0xE0))
		return 0;
	n = 10;
	*extra_off = n + 2;
	*extra_len = 0;
	*block_len = 0;
	if (flags & 0x04) {
		if (size < n + 2)
			return 0;
		int len = (int)get16(data + n);
		n += 2;
		*extra_off = n;
		*extra_len = len;
		if (size < n + len)
			return 0;
		n += len;
	}
	if (flags & 0x08) while (n < size && data[n++]) ;
	if (flags & 0x10) while (n < size && data[n++]) ;
	if (flags & 0x02) {
		if (n + 2 > size)
			return 0;
		n += 2;
	}
	const uint8_t *x = data + *extra_off;
	if (*extra_len == 6 && !memcmp(x, "BC\x02\x00", 4))
		*block_len = (long long)get16(x + 4) + 1;
	else if (*extra_len == 8 && !memcmp(x, "MZ\x04\x00", 4))
		*block_len = (long long)get32(x + 4) + n + 8;
	else if (*extra_len == 20 && !memcmp(x, "IG\x10\x00", 4))
		*block_len = (long long)((uint64_t)get32(x + 4) | ((uint64_t)get32(x + 8) << 32));
	else if (*extra_len == 8 && !memcmp(x, "IG\x04\x00", 4))
		*block_len = (long long)get32(x + 4);
	else if (*extra_len == 4 && x[3] == 0x7d)
		*block_len = (long long)(get32(x) & 0xffffff);
	else
		return 0;
	return n;
}
