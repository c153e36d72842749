/*
 * zlibutil_hip.c -- see zlibutil_hip.h.  Behaviour of zlibutil_buffer_code
 * (lib/zlibutil.c:345-408): raw mode calls func directly; rfc1950 adds the
 * 78 da header and the big-endian Adler-32 (:374-397) / checks it on decode
 * (:347-366); rfc1952 adds 1f 8b 08 00 <mtime> 02 00 and CRC32+ISIZE
 * (:379-405); the argument is returned so the function can be a pthread start
 * routine (applet/7bgzf.c:211).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "zlibutil_hip.h"

static unsigned int crc_table[256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;

static void crc_init(void)
{
	for (unsigned int i = 0; i < 256; i++) {
		unsigned int c = i;
		for (int k = 0; k < 8; k++)
			c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
		crc_table[i] = c;
	}
}

unsigned int hd_crc32(unsigned int crc, const unsigned char *buf, size_t len)
{
	/* (threads of the caller arrive here together, applet/7bgzf.c:211: a plain "ready" flag raced under TSan) */
	pthread_once(&crc_once, crc_init);
	crc = ~crc;
	while (len--)
		crc = crc_table[(crc ^ *buf++) & 0xff] ^ (crc >> 8);
	return ~crc;
}

unsigned int hd_adler32(unsigned int adler, const unsigned char *buf, size_t len)
{
	unsigned int a = adler & 0xffff, b = adler >> 16;
	while (len) {
		size_t n = len < 5552 ? len : 5552;    /* largest n with no 32-bit overflow */
		len -= n;
		while (n--) {
			a += *buf++;
			b += a;
		}
		a %= 65521u;
		b %= 65521u;
	}
	return (b << 16) | a;
}

static void w32le(unsigned char *p, unsigned int v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
static void w32be(unsigned char *p, unsigned int v) { p[3] = v; p[2] = v >> 8; p[1] = v >> 16; p[0] = v >> 24; }
static unsigned int r32be(const unsigned char *p) { return p[3] | (p[2] << 8) | (p[1] << 16) | ((unsigned int)p[0] << 24); }

hd_zlibutil_buffer *hd_zlibutil_buffer_allocate(size_t destSiz, size_t sourceSiz)
{
	hd_zlibutil_buffer *z = (hd_zlibutil_buffer *)calloc(1, sizeof(*z));
	if (!z)
		return NULL;
	z->destLen = destSiz;
	z->dest = (unsigned char *)malloc(destSiz ? destSiz : 1);
	z->sourceLen = sourceSiz;
	z->source = (unsigned char *)malloc(sourceSiz ? sourceSiz : 1);
	if (!z->dest || !z->source) {
		free(z->dest);
		free(z->source);
		free(z);
		return NULL;
	}
	return z;
}

hd_zlibutil_buffer *hd_zlibutil_buffer_code(hd_zlibutil_buffer *z)
{
	if (!z->encode) {
		if (z->rfc1950) {
			z->source += 2;
			z->sourceLen -= 6;
		}
		z->ret = ((hd_zlibutil_code_dec)z->func)(z->dest, &z->destLen, z->source, z->sourceLen);
		if (z->rfc1950) {
			z->source -= 2;
			z->sourceLen += 6;
			if (!z->ret && r32be(z->source + z->sourceLen - 4) != hd_adler32(1, z->dest, z->destLen))
				z->ret = -5; /* Z_BUF_ERROR */
		}
	} else {
		if (z->rfc1950) {
			z->dest[0] = 0x78;
			z->dest[1] = 0xda;
			z->destLen -= 6;
			z->dest += 2;
		} else if (z->rfc1952) {
			z->dest[0] = 0x1f;
			z->dest[1] = 0x8b;
			z->dest[2] = 0x08;
			z->dest[3] = 0x00;
			w32le(z->dest + 4, (unsigned int)time(NULL));
			z->dest[8] = 0x02;
			z->dest[9] = 0x00;
			z->destLen -= 18;
			z->dest += 10;
		}
		z->ret = ((hd_zlibutil_code_enc)z->func)(z->dest, &z->destLen, z->source, z->sourceLen, z->level);
		if (z->rfc1950) {
			z->dest -= 2;
			if (!z->ret) {
				w32be(z->dest + 2 + z->destLen, hd_adler32(1, z->source, z->sourceLen));
				z->destLen += 6;
			}
		} else if (z->rfc1952) {
			z->dest -= 10;
			if (!z->ret) {
				w32le(z->dest + 10 + z->destLen, hd_crc32(0, z->source, z->sourceLen));
				w32le(z->dest + 10 + z->destLen + 4, (unsigned int)z->sourceLen);
				z->destLen += 18;
			}
		}
	}
	return z;
}

hd_zlibutil_buffer *hd_zlibutil_buffer_full_flush(hd_zlibutil_buffer *z)
{
	/* the reference asserts raw mode here (applet/7dictzip.c:96) */
	if (!z->encode || z->rfc1950 || z->rfc1952 || z->func != (void *)hip_deflate) {
		z->ret = -1;
		return z;
	}
	z->func = (void *)hip_deflate_flush;
	hd_zlibutil_buffer_code(z);
	z->func = (void *)hip_deflate;
	return z;
}

void hd_zlibutil_buffer_free(hd_zlibutil_buffer *z)
{
	if (z) {
		free(z->dest);
		free(z->source);
		free(z);
	}
}
