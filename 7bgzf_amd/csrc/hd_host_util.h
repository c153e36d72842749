/*
 * hd_host_util.h -- what the container hosts (hd7dictzip, hd7razf) share: little/big
 * endian fields, the CRC-32 of a concatenation from the CRCs of its parts, a clock.
 */
#ifndef HD_HOST_UTIL_H
#define HD_HOST_UTIL_H
#include <stdint.h>
#include <stdio.h>
#include <sys/stat.h>
#include <sys/time.h>

static inline uint32_t rd16(const unsigned char *p) { return p[0] | (p[1] << 8); }
static inline uint32_t rd32(const unsigned char *p) { return rd16(p) | (rd16(p + 2) << 16); }
static inline void wr16(unsigned char *p, uint32_t v) { p[0] = v & 0xff, p[1] = (v >> 8) & 0xff; }
static inline void wr32(unsigned char *p, uint32_t v) { wr16(p, v & 0xffff), wr16(p + 2, v >> 16); }
static inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

/* ---- CRC-32 of a concatenation from the CRCs of its parts ------------------------
 * crc(A||B) = crc(A) * x^(8|B|) + crc(B) in GF(2)[x] / P, reflected bit order
 * (x^0 is bit 31). */
static inline uint32_t gf_mul(uint32_t a, uint32_t b)
{
	uint32_t p = 0;
	for (uint32_t m = 1u << 31; m; m >>= 1) {
		if (a & m)
			p ^= b;
		b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1;
	}
	return p;
}

static inline uint32_t gf_xpow(uint64_t n)
{
	uint32_t p = 1u << 31, sq = 1u << 30;
	for (; n; n >>= 1) {
		if (n & 1)
			p = gf_mul(sq, p);
		sq = gf_mul(sq, sq);
	}
	return p;
}

struct crc_fold {
	uint32_t crc;
	uint32_t op_len, op;          /* cached x^(8*op_len) */
};

static inline void crc_append(struct crc_fold *f, uint32_t crc_part, uint32_t len)
{
	if (len == 0)
		return;
	if (f->op_len != len) {
		f->op_len = len;
		f->op = gf_xpow(8ull * len);
	}
	f->crc = gf_mul(f->op, f->crc) ^ crc_part;
}

static inline double now_s(void)
{
	struct timeval tv;
	gettimeofday(&tv, NULL);
	return tv.tv_sec + tv.tv_usec * 1e-6;
}

static inline long long file_size(FILE *f)
{
	struct stat st;
	if (fstat(fileno(f), &st))
		return -1;
	return st.st_size;
}

static inline void wr32be(unsigned char *p, uint32_t v) { p[0] = v >> 24, p[1] = (v >> 16) & 0xff, p[2] = (v >> 8) & 0xff, p[3] = v & 0xff; }
static inline void wr64be(unsigned char *p, uint64_t v) { wr32be(p, (uint32_t)(v >> 32)), wr32be(p + 4, (uint32_t)v); }
static inline uint32_t rd32be(const unsigned char *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
static inline uint64_t rd64be(const unsigned char *p) { return ((uint64_t)rd32be(p) << 32) | rd32be(p + 4); }

#endif
