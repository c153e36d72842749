/*
 * stub_hipdeflate.c -- TEST ONLY.  Stand-ins for the few libhipdeflate.so entry points the C host side calls, so
 * that bgzf_hook.c (the leader / member queue, its spin-then-sleep waits, three contexts in flight) and
 * zlibutil_hip.c can run under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on a machine
 * without a GPU (the pool has no GPU sanitizers).  Members are STORED blocks framed as BGZF: enough for the
 * callers to check that every call got its own block back.  Nothing here ships; the product has no CPU codec.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "hipdeflate.h"

struct hipdeflate_lat {
	uint32_t max_blocks, in_stride, slot;
	uint8_t *in, *out;
	uint32_t *olen, *crc;
	int32_t *st;
};

static uint32_t crc32_bitwise(const uint8_t *p, size_t n)
{
	uint32_t c = 0xffffffffu;
	for (size_t i = 0; i < n; i++) {
		c ^= p[i];
		for (int k = 0; k < 8; k++)
			c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
	}
	return ~c;
}

static uint32_t stored_member(uint8_t *dst, uint32_t cap, const uint8_t *src, uint32_t n, uint32_t *crc_out)
{
	const uint32_t total = 18 + 5 + n + 8;
	if (total > cap || n > 65535)
		return 0;
	static const uint8_t hdr[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
	memcpy(dst, hdr, 16);
	dst[16] = (uint8_t)(total - 1);
	dst[17] = (uint8_t)((total - 1) >> 8);
	dst[18] = 1;
	dst[19] = (uint8_t)n;
	dst[20] = (uint8_t)(n >> 8);
	dst[21] = (uint8_t)~n;
	dst[22] = (uint8_t)(~n >> 8);
	memcpy(dst + 23, src, n);
	const uint32_t c = crc32_bitwise(src, n);
	for (int k = 0; k < 4; k++) {
		dst[23 + n + k] = (uint8_t)(c >> (8 * k));
		dst[27 + n + k] = (uint8_t)(n >> (8 * k));
	}
	*crc_out = c;
	return total;
}

hipdeflate_lat *hipdeflate_lat_open(int level, int frame, uint32_t max_blocks, uint32_t max_block_bytes)
{
	(void)level;
	(void)frame;
	hipdeflate_lat *c = (hipdeflate_lat *)calloc(1, sizeof(*c));
	c->max_blocks = max_blocks;
	c->in_stride = (max_block_bytes + 15) & ~15u;
	c->slot = 65536;
	c->in = (uint8_t *)malloc((size_t)max_blocks * c->in_stride);
	c->out = (uint8_t *)malloc((size_t)max_blocks * c->slot);
	c->olen = (uint32_t *)calloc(max_blocks, 4);
	c->crc = (uint32_t *)calloc(max_blocks, 4);
	c->st = (int32_t *)calloc(max_blocks, 4);
	return c;
}

uint8_t *hipdeflate_lat_input(hipdeflate_lat *c, uint32_t i)
{
	return i < c->max_blocks ? c->in + (size_t)i * c->in_stride : NULL;
}

int hipdeflate_lat_run(hipdeflate_lat *c, const uint32_t *in_len, uint32_t n)
{
	for (uint32_t i = 0; i < n; i++) {
		c->olen[i] = stored_member(c->out + (size_t)i * c->slot, c->slot, c->in + (size_t)i * c->in_stride, in_len[i], &c->crc[i]);
		c->st[i] = c->olen[i] ? 0 : 1;
	}
	return 0;
}

const uint8_t *hipdeflate_lat_output(hipdeflate_lat *c, uint32_t i, uint32_t *out_len, uint32_t *crc32, int32_t *status)
{
	if (out_len)
		*out_len = c->olen[i];
	if (crc32)
		*crc32 = c->crc[i];
	if (status)
		*status = c->st[i];
	return c->out + (size_t)i * c->slot;
}

void hipdeflate_lat_close(hipdeflate_lat *c)
{
	if (!c)
		return;
	free(c->in);
	free(c->out);
	free(c->olen);
	free(c->crc);
	free(c->st);
	free(c);
}

int hipdeflate_batch_deflate(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks, int level,
			     int frame, uint8_t *out, uint64_t out_stride, uint32_t out_cap, uint32_t *out_len, uint32_t *crc32,
			     int32_t *status)
{
	(void)level;
	(void)frame;
	for (uint32_t i = 0; i < nblocks; i++) {
		uint32_t c = 0;
		out_len[i] = stored_member(out + i * out_stride, out_cap < out_stride ? out_cap : (uint32_t)out_stride, in + in_off[i],
					   in_len[i], &c);
		if (crc32)
			crc32[i] = c;
		if (status)
			status[i] = out_len[i] ? 0 : 1;
	}
	return 0;
}

/* zlibutil_hip.c's codecs */
int hip_deflate(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, int level)
{
	(void)level;
	if (*destLen < sourceLen + 5 || sourceLen > 65535)
		return 1;
	dest[0] = 1;
	dest[1] = (unsigned char)sourceLen;
	dest[2] = (unsigned char)(sourceLen >> 8);
	dest[3] = (unsigned char)~sourceLen;
	dest[4] = (unsigned char)(~sourceLen >> 8);
	memcpy(dest + 5, source, sourceLen);
	*destLen = sourceLen + 5;
	return 0;
}

int hip_inflate(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen)
{
	if (sourceLen < 5 || source[0] != 1)
		return 1;
	const size_t n = source[1] | (source[2] << 8);
	if (n > *destLen || 5 + n > sourceLen)
		return 3;
	memcpy(dest, source + 5, n);
	*destLen = n;
	return 0;
}

int hip_deflate_flush(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, int level)
{
	/* stored, not final, + the flush marker */
	size_t cap = *destLen;
	if (cap < sourceLen + 10)
		return 1;
	int r = hip_deflate(dest, destLen, source, sourceLen, level);
	if (r)
		return r;
	dest[0] = 0;
	memcpy(dest + *destLen, "\x00\x00\x00\xff\xff", 5);
	*destLen += 5;
	return 0;
}

/* the device list of the real library: the stub has one "device" */
int hipdeflate_device_count(void) { return 1; }
hipdeflate_lat *hipdeflate_lat_open_on(int index, int level, int frame, uint32_t max_blocks, uint32_t max_block_bytes)
{
	return index == 0 ? hipdeflate_lat_open(level, frame, max_blocks, max_block_bytes) : NULL;
}
