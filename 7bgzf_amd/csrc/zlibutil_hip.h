/*
 * zlibutil_hip.h -- host-side mirror of the reference's codec boundary
 * (lib/zlibutil.h:28-47, lib/zlibutil.c:327-415) for the hip backend, so that a
 * caller written against zlibutil_buffer / zlibutil_buffer_code reads the same
 * with func = hip_deflate / hip_inflate.  Names carry an hd_ prefix because the
 * real zlibutil.o is linked next to this library in the drop-in build
 * (INTEGRATION.md); field layout and behaviour are the reference's.
 */
#ifndef ZLIBUTIL_HIP_H
#define ZLIBUTIL_HIP_H
#include <stddef.h>
#include "hipdeflate.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	unsigned char *dest;
	size_t destLen;
	unsigned char *source;
	size_t sourceLen;
	void *func;
	int encode;
	int level;
	int rfc1950;
	int rfc1952;
	int ret;
} hd_zlibutil_buffer;

typedef int (*hd_zlibutil_code_dec)(unsigned char *, size_t *, const unsigned char *, size_t);
typedef int (*hd_zlibutil_code_enc)(unsigned char *, size_t *, const unsigned char *, size_t, int);

hd_zlibutil_buffer *hd_zlibutil_buffer_allocate(size_t destSiz, size_t sourceSiz);
hd_zlibutil_buffer *hd_zlibutil_buffer_code(hd_zlibutil_buffer *zlibbuf);
/* zlibutil_buffer_full_flush (applet/7dictzip.c:93-126, applet/7razf.c:126-160): code the buffer, then
 * leave it in full-flush form.  With func = hip_deflate the kernel emits that form itself (no re-inflate);
 * any other func gives ret = -1, as the reference does when its inflateInit2 fails. */
hd_zlibutil_buffer *hd_zlibutil_buffer_full_flush(hd_zlibutil_buffer *zlibbuf);
void hd_zlibutil_buffer_free(hd_zlibutil_buffer *zlibbuf);

/* host checksums used by the RFC 1950 / 1952 wrappers */
unsigned int hd_crc32(unsigned int crc, const unsigned char *buf, size_t len);
unsigned int hd_adler32(unsigned int adler, const unsigned char *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif
