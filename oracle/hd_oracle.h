/*
 * hd_oracle.h -- CPU oracle for the hipdeflate hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under 7bgzf_amd/ or include/ may link,
 * import or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker.
 *
 * Parity pin: hdo_inflate / hdo_crc32 / hdo_adler32 / the framing helpers are
 * checked (tests/test_oracle_vs_ref.py) against libref.so built from the
 * reference's own sources (oracle/Makefile -> oracle/_ref/), against the 151
 * malformed streams of lib/isa-l/igzip/inflate_std_vects.h, and against the
 * golden vectors under tests/golden/ that were generated with that library.
 * hdo_deflate_twin is the serial restatement of OUR GPU encoder (there is no
 * reference byte-golden for an encoder: lib/libdeflate/libdeflate.h:75-83);
 * it is pinned by the round-trip property through the reference inflaters.
 */
#ifndef HD_ORACLE_H
#define HD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CRC-32/IEEE reflected, zlib calling convention (crc=0 to start).
 * Reference: fcrc32 = crc32_gzip_refl (lib/isa-l/crc/crc_base.c:235) or zlib
 * crc32 (lib/zlib/crc32.c:1015); used at applet/7bgzf.c:269 and
 * bgzf_compress.c:194. */
uint32_t hdo_crc32(uint32_t crc, const uint8_t *buf, size_t n);

/* Adler-32, zlib calling convention (adler=1 to start).
 * Reference: lib/zlib/adler32.c:128, used by lib/zlibutil.c:363,394. */
uint32_t hdo_adler32(uint32_t adler, const uint8_t *buf, size_t n);

/* Raw-DEFLATE inflate with the accept/reject behaviour of
 * libdeflate_deflate_decompress (lib/libdeflate/decompress_template.h:44-772,
 * lib/libdeflate/deflate_decompress.c:722-1004) as called through
 * libdeflate_inflate (lib/zlibutil.c:194-204): *destLen in = capacity, out =
 * bytes produced; stops at the first BFINAL block and ignores trailing source
 * bytes; returns HD_OK / HD_BAD_DATA / HD_INSUFFICIENT_SPACE.
 * If consumed_bits is not NULL it receives the bit length of the stream. */
int hdo_inflate(uint8_t *dest, size_t *destLen, const uint8_t *source,
		size_t sourceLen, uint64_t *consumed_bits);

/* Same, for a full-flushed chunk (no final block): success when the input is used
 * up at a block boundary (zlib_inflate / igzip_inflate as 7dictzip.c:318-323 uses them). */
int hdo_inflate_flushed(uint8_t *dest, size_t *destLen, const uint8_t *source,
			size_t sourceLen, uint64_t *consumed_bits);

/* zlibutil_buffer_full_flush (applet/7dictzip.c:93-126) applied to a finished raw
 * stream in place: BFINAL of the last block cleared, empty stored block appended.
 * *len in = stream bytes, out = new length; cap = room in `stream`. */
int hdo_full_flush(uint8_t *stream, size_t *len, size_t cap, size_t max_out);

/* Stored-only encoder, byte-identical to store_deflate (lib/zlibutil.c:302). */
int hdo_store_deflate(uint8_t *dest, size_t *destLen, const uint8_t *source,
		      size_t sourceLen);

/* Serial twin of the GPU encoder (7bgzf_amd/csrc/hd_deflate.hip).  Same
 * contract as a zlibutil_code_enc (lib/zlibutil.h:47): returns 0 and sets
 * *destLen, or non-zero when the result does not fit. */
int hdo_deflate_twin(uint8_t *dest, size_t *destLen, const uint8_t *source,
		     size_t sourceLen, int level);
/* test hook: code lengths of the twin's Huffman construction for a frequency vector */
void hdo_build_lengths(const uint32_t *freq, unsigned nsyms, unsigned maxbits, uint8_t *lens_out);

/* ... in full-flush form (HD_FRAME_RAW_FLUSH; applet/7dictzip.c:93-126) */
int hdo_deflate_twin_flush(uint8_t *dest, size_t *destLen, const uint8_t *source,
			   size_t sourceLen, int level);

/* ... in latency mode (HD_FRAME_LATENCY, include/hipdeflate.h): blocks longer than HD_LAT_SEG_BYTES(level) as
 * independent flushed segments -- what bgzf_compress, hip_deflate and hip_deflate_flush produce */
int hdo_deflate_twin_lat(uint8_t *dest, size_t *destLen, const uint8_t *source,
			 size_t sourceLen, int level);
int hdo_deflate_twin_lat_flush(uint8_t *dest, size_t *destLen, const uint8_t *source,
			       size_t sourceLen, int level);

/* BGZF member framing exactly as applet/7bgzf.c:255-272 and
 * bgzf_compress.c:191-197 write it: 18-byte header, payload, CRC32, ISIZE.
 * Returns total member size, 0 if it does not fit in 65536 or in cap. */
size_t hdo_bgzf_frame(uint8_t *dst, size_t cap, const uint8_t *payload,
		      size_t payload_len, uint32_t crc, uint32_t isize);
/* the canned 28-byte EOF member (applet/7bgzf.c:283-289, bgzf_compress.c:43-49) */
size_t hdo_bgzf_eof(uint8_t *dst, size_t cap);
/* MiGz member framing as applet/7migz.c:224-233 */
size_t hdo_migz_frame(uint8_t *dst, size_t cap, const uint8_t *payload,
		      size_t payload_len, uint32_t crc, uint32_t isize);

/* zlibutil_buffer_code's wrappers (lib/zlibutil.c:374-405) around a codec's bytes */
size_t hdo_zlib_frame(uint8_t *dst, size_t cap, const uint8_t *payload,
		      size_t payload_len, uint32_t adler);
size_t hdo_gzip_frame(uint8_t *dst, size_t cap, const uint8_t *payload,
		      size_t payload_len, uint32_t mtime, uint32_t crc, uint32_t isize);

/* Header parser with the behaviour of _read_gz_header (applet/7bgzf.c:81-131):
 * returns header length n (0 = not recognised) and the member length. */
int hdo_read_gz_header(const uint8_t *data, int size, int *extra_off,
		       int *extra_len, long long *block_len);

#ifdef __cplusplus
}
#endif
#endif
