/*
 * hd_inflate.c -- oracle: raw-DEFLATE decoder with libdeflate's verdicts.
 * TEST INFRASTRUCTURE ONLY (see hd_oracle.h).
 *
 * This is a restatement, not a transcription: it decodes canonical Huffman
 * codes by the counting method (first code of each length), one bit at a time,
 * with no decode tables -- the slowest and most obviously-correct form -- and
 * reproduces the reference decoder's ACCEPT / REJECT decisions:
 *
 *  - lib/libdeflate/decompress_template.h:75-95      BFINAL/BTYPE, BTYPE 3 bad
 *  - :101-143   HLIT/HDIST/HCLEN, precode lengths in the 16,17,18,0,8,... order
 *  - :149-232   code-length RLE; 16 with no previous length is bad; overrunning
 *               HLIT+HDIST is bad
 *  - :234-279   stored: align, LEN == ~NLEN, LEN <= input left;
 *               LEN > output left -> INSUFFICIENT_SPACE
 *  - :297-330   static code lengths 8/9/7/8 and 5
 *  - lib/libdeflate/deflate_decompress.c:799-853  overfull codes are bad;
 *               incomplete codes are bad EXCEPT the empty code and the code with
 *               a single length-1 codeword, for which both '0' and '1' decode to
 *               that symbol (symbol 0 for the empty code)
 *  - :573-577   litlen symbols 286/287 decode as length 258 (not rejected)
 *  - :623-626   offset symbols 30/31 decode as offset base 24577 + 13 bits
 *  - decompress_template.h:692-704,707  literal/match that does not fit ->
 *               INSUFFICIENT_SPACE; :724 offset > bytes produced -> BAD_DATA
 *  - deflate_decompress.c:229-249, decompress_template.h:252,744  bits past the
 *               end of input read as zeros; the stream is bad iff any of them is
 *               actually CONSUMED by the time a stored block aligns or the final
 *               block ends
 *  - decompress_template.h:735-759  stops at BFINAL; trailing input is ignored;
 *               with actual_out_nbytes_ret given (lib/zlibutil.c:201) a short
 *               output is a success.
 */
#include <stdlib.h>
#include <string.h>
#include "hd_oracle.h"
#include "../include/hipdeflate_params.h"

typedef struct {
	const uint8_t *in;
	uint64_t nbits;   /* 8 * sourceLen */
	uint64_t pos;     /* bits consumed so far */
} bits_t;

static unsigned getbit(bits_t *b)
{
	unsigned v = 0;
	if (b->pos < b->nbits)
		v = (b->in[b->pos >> 3] >> (b->pos & 7)) & 1;
	b->pos++;
	return v;
}

static unsigned getbits(bits_t *b, unsigned n)
{
	unsigned v = 0;
	for (unsigned i = 0; i < n; i++)
		v |= getbit(b) << i;
	return v;
}

/* beyond sizeof(bitbuf_t) = 8 overread bytes libdeflate gives up at once
 * (deflate_decompress.c:243-244); we use the same horizon so that a stream
 * that decodes zeros forever terminates */
static int overrun(const bits_t *b) { return b->pos > b->nbits + 64; }

typedef struct {
	uint16_t count[16];   /* codewords per length */
	uint16_t sorted[288]; /* symbols ordered by (length, symbol) */
	int degenerate;       /* 1: every bit pattern decodes to sorted[0] in 1 bit */
} code_t;

/* returns 0 if the lengths are rejected by build_decode_table() */
static int build_code(code_t *c, const uint8_t *lens, unsigned nsyms, unsigned maxlen)
{
	unsigned offs[17];
	memset(c, 0, sizeof(*c));
	for (unsigned s = 0; s < nsyms; s++)
		c->count[lens[s]]++;
	while (maxlen > 1 && c->count[maxlen] == 0)
		maxlen--;
	uint32_t used = 0;
	for (unsigned l = 1; l <= maxlen; l++)
		used = (used << 1) + c->count[l];
	offs[1] = 0;
	for (unsigned l = 1; l < 16; l++)
		offs[l + 1] = offs[l] + c->count[l];
	for (unsigned s = 0; s < nsyms; s++)
		if (lens[s])
			c->sorted[offs[lens[s]]++] = (uint16_t)s;
	if (used > (1u << maxlen))
		return 0;                         /* overfull */
	if (used < (1u << maxlen)) {              /* incomplete */
		if (used == 0) {
			c->sorted[0] = 0;
		} else if (used != (1u << (maxlen - 1)) || c->count[1] != 1) {
			return 0;
		}
		c->degenerate = 1;
	}
	c->count[0] = 0;
	return 1;
}

static unsigned decode_sym(bits_t *b, const code_t *c)
{
	if (c->degenerate) {
		(void)getbit(b);
		return c->sorted[0];
	}
	/* canonical code: codewords of one length are consecutive integers, and
	 * the first codeword of length l+1 is (first_l + count_l) << 1 */
	unsigned code = 0, first = 0, index = 0;
	for (unsigned l = 1; l <= 15; l++) {
		code |= getbit(b);
		unsigned cnt = c->count[l];
		if (code - first < cnt)
			return c->sorted[index + (code - first)];
		index += cnt;
		first = (first + cnt) << 1;
		code <<= 1;
	}
	return 0xffff; /* unreachable for a complete code */
}

static const uint16_t len_base[31]  = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
	35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 258, 258 };
static const uint8_t  len_extra[31] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
	3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0 };
static const uint16_t off_base[32]  = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
	257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577,
	24577, 24577 };
static const uint8_t  off_extra[32] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
	7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 13, 13 };
static const uint8_t  precode_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3,
	13, 2, 14, 1, 15 };

static int inflate_ex(uint8_t *dest, size_t *destLen, const uint8_t *source,
		      size_t sourceLen, uint64_t *consumed_bits, uint64_t *last_header_bit, int flushed);

int hdo_inflate(uint8_t *dest, size_t *destLen, const uint8_t *source,
		size_t sourceLen, uint64_t *consumed_bits)
{
	return inflate_ex(dest, destLen, source, sourceLen, consumed_bits, NULL, 0);
}

/* The inflate the reference's 7dictzip / 7razf readers get from zlib_inflate
 * (lib/zlibutil.c:266-300: Z_BUF_ERROR, i.e. out of input, ends the loop and is
 * success) and igzip_inflate (lib/zlibutil_igzip.c:93-119: ISAL_END_INPUT is
 * success) on a full-flushed chunk, which has no final block: the stream may stop
 * after a non-final block when every input byte has been used.  Stricter than the
 * reference on purpose -- input that runs out INSIDE a block stays HD_BAD_DATA. */
int hdo_inflate_flushed(uint8_t *dest, size_t *destLen, const uint8_t *source,
			size_t sourceLen, uint64_t *consumed_bits)
{
	return inflate_ex(dest, destLen, source, sourceLen, consumed_bits, NULL, 1);
}

/* Restatement of zlibutil_buffer_full_flush (applet/7dictzip.c:93-126,
 * applet/7razf.c:126-160) for a finished raw-DEFLATE stream: the reference
 * re-inflates it with a zlib whose inflate() clears the BFINAL bit of the last
 * block header in place and reports how many bits of the last byte are unused
 * (7razf_testdecode.c:595-606,1024); fewer than 3 -> one more zero byte (the
 * empty stored block's header needs 3), then 00 00 ff ff.  max_out bounds the
 * scratch the stream is inflated into.  Returns 0, or the inflate verdict. */
int hdo_full_flush(uint8_t *stream, size_t *len, size_t cap, size_t max_out)
{
	uint64_t end = 0, hdr = 0;
	size_t outlen = max_out;
	uint8_t *tmp = malloc(max_out ? max_out : 1);
	int r = inflate_ex(tmp, &outlen, stream, *len, &end, &hdr, 0);
	free(tmp);
	if (r)
		return r;
	size_t n = (size_t)((end + 7) >> 3);
	unsigned unused = (unsigned)(8 * n - end);
	if (n + (unused < 3) + 4 > cap)
		return HD_INSUFFICIENT_SPACE;
	stream[hdr >> 3] &= (uint8_t)~(1u << (hdr & 7));
	if (unused < 3)
		stream[n++] = 0;
	stream[n++] = 0; stream[n++] = 0; stream[n++] = 0xff; stream[n++] = 0xff;
	*len = n;
	return 0;
}

static int inflate_ex(uint8_t *dest, size_t *destLen, const uint8_t *source,
		      size_t sourceLen, uint64_t *consumed_bits, uint64_t *last_header_bit, int flushed)
{
	bits_t b = { source, (uint64_t)sourceLen * 8, 0 };
	size_t cap = *destLen, out = 0;
	code_t pre, lit, off;
	uint8_t lens[288 + 32 + 138];

	for (;;) {
		if (last_header_bit)
			*last_header_bit = b.pos;
		unsigned bfinal = getbit(&b);
		unsigned btype = getbits(&b, 2);

		if (btype == 0) {
			b.pos = (b.pos + 7) & ~(uint64_t)7;
			if (b.pos > b.nbits)
				return HD_BAD_DATA;
			size_t ip = (size_t)(b.pos >> 3);
			if (sourceLen - ip < 4)
				return HD_BAD_DATA;
			unsigned len = source[ip] | (source[ip + 1] << 8);
			unsigned nlen = source[ip + 2] | (source[ip + 3] << 8);
			ip += 4;
			if (len != (~nlen & 0xffff))
				return HD_BAD_DATA;
			if (len > cap - out)
				return HD_INSUFFICIENT_SPACE;
			if (len > sourceLen - ip)
				return HD_BAD_DATA;
			memcpy(dest + out, source + ip, len);
			out += len;
			b.pos = (uint64_t)(ip + len) * 8;
		} else {
			unsigned nlit, noff;
			if (btype == 2) {
				nlit = 257 + getbits(&b, 5);
				noff = 1 + getbits(&b, 5);
				unsigned npre = 4 + getbits(&b, 4);
				uint8_t plens[19] = { 0 };
				for (unsigned i = 0; i < npre; i++)
					plens[precode_order[i]] = (uint8_t)getbits(&b, 3);
				if (!build_code(&pre, plens, 19, 7))
					return HD_BAD_DATA;
				unsigned i = 0;
				while (i < nlit + noff) {
					if (overrun(&b))
						return HD_BAD_DATA;
					unsigned s = decode_sym(&b, &pre);
					if (s < 16) {
						lens[i++] = (uint8_t)s;
					} else if (s == 16) {
						if (i == 0)
							return HD_BAD_DATA;
						unsigned rep = 3 + getbits(&b, 2);
						memset(lens + i, lens[i - 1], rep);
						i += rep;
					} else if (s == 17) {
						unsigned rep = 3 + getbits(&b, 3);
						memset(lens + i, 0, rep);
						i += rep;
					} else {
						unsigned rep = 11 + getbits(&b, 7);
						memset(lens + i, 0, rep);
						i += rep;
					}
				}
				if (i != nlit + noff)
					return HD_BAD_DATA;
			} else if (btype == 1) {
				unsigned i = 0;
				for (; i < 144; i++) lens[i] = 8;
				for (; i < 256; i++) lens[i] = 9;
				for (; i < 280; i++) lens[i] = 7;
				for (; i < 288; i++) lens[i] = 8;
				for (; i < 320; i++) lens[i] = 5;
				nlit = 288;
				noff = 32;
			} else {
				return HD_BAD_DATA;
			}
			/* offset code first, as decompress_template.h:335-336 */
			if (!build_code(&off, lens + nlit, noff, 15))
				return HD_BAD_DATA;
			if (!build_code(&lit, lens, nlit, 15))
				return HD_BAD_DATA;

			for (;;) {
				if (overrun(&b))
					return HD_BAD_DATA;
				unsigned s = decode_sym(&b, &lit);
				if (s < 256) {
					if (out == cap)
						return HD_INSUFFICIENT_SPACE;
					dest[out++] = (uint8_t)s;
					continue;
				}
				if (s == 256)
					break;
				s -= 257;
				unsigned length = len_base[s] + getbits(&b, len_extra[s]);
				if (length > cap - out)
					return HD_INSUFFICIENT_SPACE;
				unsigned os = decode_sym(&b, &off);
				unsigned offset = off_base[os] + getbits(&b, off_extra[os]);
				if (offset > out)
					return HD_BAD_DATA;
				for (unsigned k = 0; k < length; k++, out++)
					dest[out] = dest[out - offset];
			}
		}
		if (bfinal)
			break;
		if (flushed && b.pos <= b.nbits && b.pos + 7 >= b.nbits)
			break;
		if (overrun(&b))
			return HD_BAD_DATA;
	}
	if (b.pos > b.nbits)
		return HD_BAD_DATA;
	if (consumed_bits)
		*consumed_bits = b.pos;
	*destLen = out;
	return HD_OK;
}
