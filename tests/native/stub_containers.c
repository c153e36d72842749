/*
 * stub_containers.c -- TEST ONLY.  Stand-ins for the libhipdeflate.so entry points the seven container hosts call
 * (hd_{bgzf,dictzip,razf,gzinga,ciso,daxcr,png}_host.c), so that their PARSERS of untrusted files -- offset tables, member
 * lengths, index members, PNG chunk lengths -- can run under AddressSanitizer / UndefinedBehaviorSanitizer on a machine
 * without a GPU (the pool has no GPU sanitizers).  The "codec" writes STORED blocks in every frame of include/hipdeflate.h
 * and inflates stored blocks only; anything else is HD_BAD_DATA.  Nothing here ships; the product has no CPU codec.
 * Driven by tests/native/container_fuzz.c (tests/test_sanitize_hosts.py).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hipdeflate.h"
#include "hipdeflate_params.h"

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

static uint32_t adler32_simple(const uint8_t *p, size_t n)
{
	uint32_t a = 1, b = 0;
	for (size_t i = 0; i < n; i++) {
		a = (a + p[i]) % 65521u;
		b = (b + a) % 65521u;
	}
	return (b << 16) | a;
}

int hipdeflate_init(int device) { (void)device; return 0; }
int hipdeflate_available(void) { return 0; }
void hipdeflate_shutdown(void) {}
const char *hipdeflate_version(void) { return "hipdeflate TEST STUB (stored blocks on the CPU; never shipped)"; }

uint64_t hipdeflate_bound(uint64_t n, int level)
{
	(void)level;
	return (n + 5 * (n / 65535 + 1) + 5 + 32 + 15) & ~(uint64_t)15;
}

/* raw stored stream of src[0..n): returns bytes, 0 if cap is too small */
static size_t stored_raw(uint8_t *dst, size_t cap, const uint8_t *src, size_t n, int flush)
{
	const size_t need = HD_STORED_SIZE(n) + (flush ? 5u : 0u);
	if (need > cap)
		return 0;
	size_t o = 0, left = n;
	do {
		const size_t blk = left < 65535 ? left : 65535;
		dst[o] = (left - blk || flush) ? 0 : 1;
		dst[o + 1] = (uint8_t)blk;
		dst[o + 2] = (uint8_t)(blk >> 8);
		dst[o + 3] = (uint8_t)~blk;
		dst[o + 4] = (uint8_t)(~blk >> 8);
		if (blk)
			memcpy(dst + o + 5, src + (n - left), blk);
		o += 5 + blk;
		left -= blk;
	} while (left);
	if (flush) {
		memcpy(dst + o, "\x00\x00\x00\xff\xff", 5);
		o += 5;
	}
	return o;
}

static uint32_t member(int frame, uint8_t *dst, size_t cap, const uint8_t *src, uint32_t n, uint32_t *crc_out)
{
	frame &= ~HD_FRAME_LATENCY;
	const uint32_t hdr = frame == HD_FRAME_BGZF ? 18 : frame == HD_FRAME_MIGZ ? 20 : frame == HD_FRAME_ZLIB ? 2 : frame == HD_FRAME_GZIP ? 10 : 0;
	const uint32_t trl = frame == HD_FRAME_ZLIB ? 4 : (frame == HD_FRAME_BGZF || frame == HD_FRAME_MIGZ || frame == HD_FRAME_GZIP) ? 8 : 0;
	if (frame == HD_FRAME_BGZF && cap > 65536)
		cap = 65536;
	if (cap < hdr + trl)
		return 0;
	const size_t pay = stored_raw(dst + hdr, cap - hdr - trl, src, n, frame == HD_FRAME_RAW_FLUSH);
	if (!pay)
		return 0;
	const uint32_t crc = crc32_bitwise(src, n), total = hdr + (uint32_t)pay + trl;
	*crc_out = crc;
	if (frame == HD_FRAME_BGZF || frame == HD_FRAME_MIGZ) {
		static const uint8_t h[10] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff };
		memcpy(dst, h, 10);
		if (frame == HD_FRAME_BGZF) {
			memcpy(dst + 10, "\x06\x00" "BC" "\x02\x00", 6);
			dst[16] = (uint8_t)(total - 1);
			dst[17] = (uint8_t)((total - 1) >> 8);
		} else {
			memcpy(dst + 10, "\x08\x00" "MZ" "\x04\x00", 6);
			for (int k = 0; k < 4; k++)
				dst[16 + k] = (uint8_t)(pay >> (8 * k));
		}
	} else if (frame == HD_FRAME_GZIP) {
		static const uint8_t h[10] = { 0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 0 };
		memcpy(dst, h, 10);
	} else if (frame == HD_FRAME_ZLIB) {
		dst[0] = 0x78;
		dst[1] = 0xda;
	}
	uint8_t *t = dst + hdr + pay;
	if (frame == HD_FRAME_ZLIB) {
		const uint32_t a = adler32_simple(src, n);
		for (int k = 0; k < 4; k++)
			t[k] = (uint8_t)(a >> (8 * (3 - k)));
	} else if (trl) {
		for (int k = 0; k < 4; k++) {
			t[k] = (uint8_t)(crc >> (8 * k));
			t[4 + k] = (uint8_t)(n >> (8 * k));
		}
	}
	return total;
}

int hipdeflate_batch_deflate(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks, int level,
			     int frame, uint8_t *out, uint64_t out_stride, uint32_t out_cap, uint32_t *out_len, uint32_t *crc32,
			     int32_t *status)
{
	(void)level;
	for (uint32_t i = 0; i < nblocks; i++) {
		uint32_t c = 0;
		out_len[i] = member(frame, out + i * out_stride, out_cap < out_stride ? out_cap : (size_t)out_stride, in + in_off[i],
				    in_len[i], &c);
		if (crc32)
			crc32[i] = c;
		if (status)
			status[i] = out_len[i] ? 0 : 1;
	}
	return 0;
}

/* stored blocks only; trailing source bytes allowed (the callers hand payload + trailer) */
static int inflate_stored(uint8_t *dst, size_t cap, const uint8_t *src, size_t n, size_t *out_n, int flushed)
{
	size_t ip = 0, op = 0;
	for (;;) {
		if (n - ip >= 2 && src[ip] == 3 && src[ip + 1] == 0)
			break;
		if (n - ip < 5)
			return (flushed && ip == n && ip) ? (*out_n = op, HD_OK) : HD_BAD_DATA;
		const unsigned hdr = src[ip];
		if (hdr == 3 && src[ip + 1] == 0)
			break;                                        /* 03 00: the empty final block of an EOF member / a segmented stream */
		if (hdr & 6)
			return HD_BAD_DATA;                           /* not a stored block: this stub knows nothing else */
		const size_t len = src[ip + 1] | (src[ip + 2] << 8), nlen = src[ip + 3] | (src[ip + 4] << 8);
		if (len != (~nlen & 0xffff))
			return HD_BAD_DATA;
		ip += 5;
		if (len > n - ip)
			return HD_BAD_DATA;
		if (len > cap - op)
			return HD_INSUFFICIENT_SPACE;
		memcpy(dst + op, src + ip, len);
		ip += len;
		op += len;
		if (hdr & 1)
			break;
		if (flushed && ip == n)
			break;
	}
	*out_n = op;
	return HD_OK;
}

static int batch_inflate(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks, uint8_t *out,
			 const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, uint32_t *crc32, int32_t *status,
			 int flushed)
{
	for (uint32_t i = 0; i < nblocks; i++) {
		if (in_len[i] >= HD_INFLATE_MAX_IN)
			return HD_E_ARG;
		size_t n = 0;
		const int r = inflate_stored(out + out_off[i], out_cap[i], in + in_off[i], in_len[i], &n, flushed);
		out_len[i] = r ? 0 : (uint32_t)n;
		if (crc32)
			crc32[i] = r ? 0 : crc32_bitwise(out + out_off[i], n);
		if (status)
			status[i] = r;
	}
	return 0;
}

int hipdeflate_batch_inflate(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks, uint8_t *out,
			     const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len, uint32_t *crc32, int32_t *status)
{
	return batch_inflate(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, crc32, status, 0);
}

int hipdeflate_batch_inflate_flush(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks,
				   uint8_t *out, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
				   uint32_t *crc32, int32_t *status)
{
	return batch_inflate(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, crc32, status, 1);
}

int hip_deflate(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, int level)
{
	(void)level;
	const size_t n = stored_raw(dest, *destLen, source, sourceLen, 0);
	if (!n)
		return 1;
	*destLen = n;
	return 0;
}

int hip_deflate_flush(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, int level)
{
	(void)level;
	const size_t n = stored_raw(dest, *destLen, source, sourceLen, 1);
	if (!n)
		return 1;
	*destLen = n;
	return 0;
}

int hip_inflate(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen)
{
	size_t n = 0;
	const int r = inflate_stored(dest, *destLen, source, sourceLen, &n, 0);
	if (!r)
		*destLen = n;
	return r;
}

int hip_inflate_flush(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen)
{
	size_t n = 0;
	const int r = inflate_stored(dest, *destLen, source, sourceLen, &n, 1);
	if (!r)
		*destLen = n;
	return r;
}

/* ---- the streaming pipes of hd_bgzf_host.c: `depth` slots, the work done inside submit(), the calling protocol of
 * the real ones (input() waits for a free slot, result() hands out the oldest submitted batch and frees the one before) */
#include <pthread.h>
typedef struct {
	uint8_t *in, *out;
	uint32_t *olen, *crc;
	uint64_t *doff;
	size_t nbytes, total;
	uint32_t nb;
	int verdict, state;      /* 0 free, 1 input handed out, 2 submitted, 3 result held */
} stub_slot;

struct hipdeflate_pipe {
	int level, frame, depth;
	uint32_t block, per_batch;
	size_t slot;
	stub_slot s[16];
	pthread_mutex_t mu;
	pthread_cond_t cv;
	uint64_t n_in, n_sub, n_out;
	int held;
};

hipdeflate_pipe *hipdeflate_pipe_open(int level, int frame, uint32_t block_bytes, uint32_t blocks_per_batch, int depth)
{
	if (frame < HD_FRAME_RAW || frame > HD_FRAME_GZIP || !block_bytes || (block_bytes & 15) || !blocks_per_batch || depth < 2 || depth > 16)
		return NULL;
	hipdeflate_pipe *p = (hipdeflate_pipe *)calloc(1, sizeof(*p));
	p->level = level;
	p->frame = frame;
	p->depth = depth;
	p->block = block_bytes;
	p->per_batch = blocks_per_batch;
	p->slot = (size_t)hipdeflate_bound(block_bytes, level);
	p->held = -1;
	pthread_mutex_init(&p->mu, NULL);
	pthread_cond_init(&p->cv, NULL);
	for (int k = 0; k < depth; k++) {
		p->s[k].in = (uint8_t *)malloc((size_t)block_bytes * blocks_per_batch);
		p->s[k].out = (uint8_t *)malloc(p->slot * blocks_per_batch);
		p->s[k].olen = (uint32_t *)calloc(blocks_per_batch, 4);
		p->s[k].crc = (uint32_t *)calloc(blocks_per_batch, 4);
		p->s[k].doff = (uint64_t *)calloc(blocks_per_batch, 8);
	}
	return p;
}

uint8_t *hipdeflate_pipe_input(hipdeflate_pipe *p, size_t *cap)
{
	if (!p)
		return NULL;
	pthread_mutex_lock(&p->mu);
	stub_slot *s = &p->s[p->n_in % p->depth];
	if (s->state == 1) {
		pthread_mutex_unlock(&p->mu);
		return NULL;
	}
	while (s->state != 0)
		pthread_cond_wait(&p->cv, &p->mu);
	s->state = 1;
	pthread_mutex_unlock(&p->mu);
	if (cap)
		*cap = (size_t)p->block * p->per_batch;
	return s->in;
}

int hipdeflate_pipe_submit(hipdeflate_pipe *p, size_t nbytes)
{
	if (!p)
		return HD_E_ARG;
	stub_slot *s = &p->s[p->n_in % p->depth];
	if (s->state != 1 || nbytes > (size_t)p->block * p->per_batch)
		return HD_E_ARG;
	s->nbytes = nbytes;
	s->nb = (uint32_t)((nbytes + p->block - 1) / p->block);
	s->total = 0;
	s->verdict = 0;
	for (uint32_t i = 0; i < s->nb; i++) {
		const uint32_t len = i + 1 < s->nb ? p->block : (uint32_t)(nbytes - (size_t)i * p->block);
		s->doff[i] = s->total;
		s->olen[i] = member(p->frame, s->out + s->total, p->slot, s->in + (size_t)i * p->block, len, &s->crc[i]);
		if (!s->olen[i])
			s->verdict = 1;
		s->total += s->olen[i];
	}
	pthread_mutex_lock(&p->mu);
	s->state = 2;
	p->n_in++;
	p->n_sub++;
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
	return 0;
}

int hipdeflate_pipe_result(hipdeflate_pipe *p, const uint8_t **data, size_t *nbytes, uint32_t *nblocks)
{
	if (!p || !data || !nbytes)
		return HD_E_ARG;
	pthread_mutex_lock(&p->mu);
	if (p->held >= 0) {
		p->s[p->held].state = 0;
		p->held = -1;
		pthread_cond_broadcast(&p->cv);
	}
	if (p->n_out == p->n_sub) {
		pthread_mutex_unlock(&p->mu);
		return HD_E_ARG;
	}
	stub_slot *s = &p->s[p->n_out % p->depth];
	*data = s->out;
	*nbytes = s->total;
	if (nblocks)
		*nblocks = s->nb;
	s->state = 3;
	p->held = (int)(p->n_out % p->depth);
	p->n_out++;
	pthread_mutex_unlock(&p->mu);
	return s->verdict;
}

int hipdeflate_pipe_members(hipdeflate_pipe *p, const uint32_t **out_len, const uint64_t **dst_off, const uint32_t **crc32)
{
	if (!p || p->held < 0)
		return HD_E_ARG;
	const stub_slot *s = &p->s[p->held];
	if (out_len)
		*out_len = s->nb ? s->olen : NULL;
	if (dst_off)
		*dst_off = s->nb ? s->doff : NULL;
	if (crc32)
		*crc32 = s->nb ? s->crc : NULL;
	return 0;
}

void hipdeflate_pipe_close(hipdeflate_pipe *p)
{
	if (!p)
		return;
	for (int k = 0; k < p->depth; k++) {
		free(p->s[k].in);
		free(p->s[k].out);
		free(p->s[k].olen);
		free(p->s[k].crc);
		free(p->s[k].doff);
	}
	free(p);
}

struct hipdeflate_unpipe {
	int depth;
	uint32_t max_members;
	size_t in_cap, out_cap;
	stub_slot s[16];
	pthread_mutex_t mu;
	pthread_cond_t cv;
	uint64_t n_in, n_sub, n_out;
	int held;
};

hipdeflate_unpipe *hipdeflate_unpipe_open(uint32_t max_members, size_t in_cap, size_t out_cap, int depth)
{
	if (!max_members || !in_cap || !out_cap || depth < 2 || depth > 16)
		return NULL;
	hipdeflate_unpipe *p = (hipdeflate_unpipe *)calloc(1, sizeof(*p));
	p->depth = depth;
	p->max_members = max_members;
	p->in_cap = in_cap;
	p->out_cap = out_cap;
	p->held = -1;
	pthread_mutex_init(&p->mu, NULL);
	pthread_cond_init(&p->cv, NULL);
	for (int k = 0; k < depth; k++) {
		p->s[k].in = (uint8_t *)malloc(in_cap);
		p->s[k].out = (uint8_t *)malloc(out_cap);
	}
	return p;
}

uint8_t *hipdeflate_unpipe_input(hipdeflate_unpipe *p, size_t *cap)
{
	if (!p)
		return NULL;
	pthread_mutex_lock(&p->mu);
	stub_slot *s = &p->s[p->n_in % p->depth];
	if (s->state == 1) {
		pthread_mutex_unlock(&p->mu);
		return NULL;
	}
	while (s->state != 0)
		pthread_cond_wait(&p->cv, &p->mu);
	s->state = 1;
	pthread_mutex_unlock(&p->mu);
	if (cap)
		*cap = p->in_cap;
	return s->in;
}

int hipdeflate_unpipe_submit(hipdeflate_unpipe *p, const uint64_t *in_off, const uint32_t *in_len, const uint32_t *out_size,
			     uint32_t nmembers)
{
	if (!p || nmembers > p->max_members || (nmembers && (!in_off || !in_len || !out_size)))
		return HD_E_ARG;
	stub_slot *s = &p->s[p->n_in % p->depth];
	if (s->state != 1)
		return HD_E_ARG;
	size_t osum = 0, in_end = 0;
	for (uint32_t i = 0; i < nmembers; i++) {
		if (in_len[i] >= HD_INFLATE_MAX_IN)
			return HD_E_ARG;
		osum += out_size[i];
		if (in_off[i] + in_len[i] > in_end)
			in_end = (size_t)(in_off[i] + in_len[i]);
	}
	if (in_end > p->in_cap || osum > p->out_cap)
		return HD_E_ARG;
	s->verdict = 0;
	osum = 0;
	for (uint32_t i = 0; i < nmembers; i++) {
		size_t n = 0;
		const int r = inflate_stored(s->out + osum, out_size[i], s->in + in_off[i], in_len[i], &n, 0);
		if (!s->verdict)
			s->verdict = r ? r : (n != out_size[i] ? HD_INSUFFICIENT_SPACE : 0);
		osum += out_size[i];
	}
	s->nbytes = osum;
	pthread_mutex_lock(&p->mu);
	s->state = 2;
	p->n_in++;
	p->n_sub++;
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
	return 0;
}

int hipdeflate_unpipe_result(hipdeflate_unpipe *p, const uint8_t **data, size_t *nbytes)
{
	if (!p || !data || !nbytes)
		return HD_E_ARG;
	pthread_mutex_lock(&p->mu);
	if (p->held >= 0) {
		p->s[p->held].state = 0;
		p->held = -1;
		pthread_cond_broadcast(&p->cv);
	}
	if (p->n_out == p->n_sub) {
		pthread_mutex_unlock(&p->mu);
		return HD_E_ARG;
	}
	stub_slot *s = &p->s[p->n_out % p->depth];
	*data = s->out;
	*nbytes = s->nbytes;
	s->state = 3;
	p->held = (int)(p->n_out % p->depth);
	p->n_out++;
	pthread_mutex_unlock(&p->mu);
	return s->verdict;
}

void hipdeflate_unpipe_close(hipdeflate_unpipe *p)
{
	if (!p)
		return;
	for (int k = 0; k < p->depth; k++) {
		free(p->s[k].in);
		free(p->s[k].out);
	}
	free(p);
}

/* the device list of the real library: the stub answers for HIPDEFLATE_DEVICES-many "devices" (default one), all the
 * same CPU code -- enough for hd7bgzf -g N to deal its batches round robin under the sanitizers */
static int stub_ndev(void)
{
	const char *s = getenv("HIPDEFLATE_DEVICES");
	int n = 1;
	for (; s && *s; s++)
		n += *s == ',';
	return n;
}
static int g_stub_ndev;
int hipdeflate_device_count(void) { return g_stub_ndev ? g_stub_ndev : stub_ndev(); }
int hipdeflate_init_devices(const int *devices, int n) { (void)devices; g_stub_ndev = n; return n >= 1 ? 0 : HD_E_ARG; }
int hipdeflate_use_device(int index) { return index >= 0 && index < hipdeflate_device_count() ? 0 : HD_E_ARG; }
hipdeflate_pipe *hipdeflate_pipe_open_on(int index, int level, int frame, uint32_t block_bytes, uint32_t blocks_per_batch, int depth)
{
	return index >= 0 && index < hipdeflate_device_count() ? hipdeflate_pipe_open(level, frame, block_bytes, blocks_per_batch, depth) : NULL;
}
hipdeflate_unpipe *hipdeflate_unpipe_open_on(int index, uint32_t max_members, size_t in_cap, size_t out_cap, int depth)
{
	return index >= 0 && index < hipdeflate_device_count() ? hipdeflate_unpipe_open(max_members, in_cap, out_cap, depth) : NULL;
}
