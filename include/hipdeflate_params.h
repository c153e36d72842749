/*
 * hipdeflate_params.h -- algorithm geometry shared by the HIP kernels
 * (7bgzf_amd/csrc) and their CPU twin (oracle/hd_deflate_twin.c).
 *
 * The GPU encoder is NOT libdeflate: it is a wave64-native parse (64 input
 * positions looked up at once, greedy resolution by ballot, bit packing by a
 * wave prefix scan).  Its output is valid RFC 1951; it is byte-identical to
 * the CPU twin, which restates the same algorithm serially.  Everything both
 * sides must agree on lives here so that a change of geometry cannot make
 * them drift apart silently.
 *
 * Level map (role of lib/libdeflate/deflate_compress.c:3874-3990 in the
 * reference, where a level selects matchfinder + parser + Huffman mode):
 *   0      stored blocks only          (store_deflate, lib/zlibutil.c:302)
 *   1      greedy parse, static Huffman (BASELINE config 2, "level-1-like"), one wavefront's 4 KiB window
 *   2      greedy parse, dynamic Huffman, the same window (fast on DNA-like data)
 *   3..9   the workgroup parse -- a 32 KiB window and a 64 KiB multi-way table shared by the sixteen wavefronts of a
 *          workgroup, block splitting ("WORKGROUP LEVELS" below; config 5, "level-6-like"): 3 one way greedy, 4 one way
 *          lazy, 5 two ways, 6..9 four ways (7..9 = 6: the plateau is measured).  ONE codec per level since
 *          round 5: HD_FRAME_LATENCY (one block per call, the hook) changes the schedule of these levels -- more
 *          wavefronts on one block -- not their bytes.  Levels 1..2 keep a latency FORM of their own (segments, parts:
 *          HD_LAT_* below).
 */
#ifndef HIPDEFLATE_PARAMS_H
#define HIPDEFLATE_PARAMS_H

#define HD_WAVE            64          /* positions parsed per step = lanes */
#define HD_MIN_MATCH       4           /* hash covers 4 bytes               */
#define HD_MAX_MATCH       258         /* RFC 1951 3.2.5                    */
#define HD_PIECE           1024        /* window refill granule (64 x 16 B) */
#define HD_LOOKAHEAD       384         /* >= 64 + 258 + 8, bytes past S     */

/* level 1: static Huffman, streaming emit (no token buffer) */
#ifndef HD_L1_WIN_BITS                 /* (overridable for tools/r05_l1_geometry.sh: what a wider window would cost and buy) */
#define HD_L1_WIN_BITS     12          /* 4 KiB LDS ring window (occupancy)  */
#define HD_L1_HASH_BITS    11          /* hash bits; 1536 x u16 entries kept (HD_TABLE_ENTRIES) */
#endif

/* level 2: greedy parse, dynamic Huffman, in the level-1 geometry (4 KiB ring, 2^11 table).
 * On DNA-like data the window size hardly matters (same ratio as level 3), and the small LDS
 * footprint lets the parse run at level-1 occupancy: blocks <= 64 KiB go through a parse kernel
 * and an emit kernel (hd_deflate_static.hpp / hd_deflate_dynamic.hpp), larger ones through the
 * fused kernel; the bytes are the same either way */
#define HD_L2_WIN_BITS     12
#define HD_L2_HASH_BITS    11
#define HD_L2_MIN_LEN      4
/* (rounds 1-4 had one-wavefront geometries for levels 3..9 here -- 8 / 16 / 32 KiB rings, one- and two-way tables: HD_L3.. / HD_L9..;
 * round 5 removed them with the last kernels that used them: those levels are the workgroup parse in every form, docs/rounds.md) */

/* Hash of the four bytes v -> table slot, in 24-bit multiplies (full rate on CDNA; a 32-bit v_mul_lo is
 * quarter rate) and scaled to ANY table size without a power-of-two step:
 *     t    = v[23:0] * K1 + v[31:16] * K2          (mod 2^32: v_mul_u32_u24 + v_mad_u32_u24)
 *     slot = ((t >> 16) * entries) >> 16            (v_mul_u32_u24 on the high word)
 * Four VALU instructions with SDWA word selects (hd_device.hpp hash_slot_addr) where the round-1 hash
 * (v * 0x9E3779B1 >> shift, then * 3 / 4) cost eight issue slots -- and the FASTQ-like set comes out 1.5 %
 * smaller at level 1 (0.4578 -> 0.4508), text the same: the uniform scaling wastes no slot. */
#define HD_HASH_K1         0x9E3779u
#define HD_HASH_K2         0xC2B2AEu
#define HD_HASH_SLOT(v, entries) \
	((((((uint32_t)(v) & 0xffffffu) * HD_HASH_K1 + ((uint32_t)(v) >> 16) * HD_HASH_K2) >> 16) * (uint32_t)(entries)) >> 16)
/* LAZY LEVELS (6..9; level 5 keeps the one-way table: the faster step of the ladder): the hash covers SIX bytes and a
 * bucket holds TWO positions.
 *   key      v = bytes [p, p+4), vh = bytes [p+4, p+6):  t += vh * K3 in the sum above (one more 24-bit multiply).  A
 *            position enters the table (and looks into it) only with six bytes left.  DNA-like data has 256 distinct
 *            4-byte keys in its reads; six bytes find the far repeats libdeflate's chains find (level 6, FASTQ-like
 *            set: 0.285 -> 0.278 of the input with this alone).
 *   bucket   one dword: low half the newest position of the key, high half the one before it.  A step reads the
 *            bucket and stores (bucket << 16) | own position -- the same one LDS read and one LDS write per lane as
 *            the one-way table; of the lanes of a step that share a bucket the highest keeps its store (as before),
 *            so the bucket becomes { newest before the step, highest lane of the step }.
 *   choice   both candidates are verified over 16 bytes; the older one is taken only when it is strictly longer.
 * The role of hc_matchfinder's chain walk (lib/libdeflate/hc_matchfinder.h:183-338: depth 35 at level 6) with the
 * depth LDS affords: 2.  tests/golden/ratio_ref.json + hdtest.RATIO_BOUNDS hold the resulting sizes against libdeflate's. */
#define HD_HASH_K3         0x85EBCAu
#define HD_HASH_SLOT6(v, vh, entries) \
	((((((uint32_t)(v) & 0xffffffu) * HD_HASH_K1 + ((uint32_t)(v) >> 16) * HD_HASH_K2 + ((uint32_t)(vh) & 0xffffu) * HD_HASH_K3) >> 16) * (uint32_t)(entries)) >> 16)
#define HD_DEEP_LEVEL      6           /* first level with the two-way buckets */
#define HD_LAZY_KEY_BYTES  6
#define HD_LAZY_WAYS       2
/* buckets of the two-way tables: 2560 x 4 B with the 8 KiB ring of levels 5..6 (8 parse waves per CU), 2560 / 4096 with the
 * 16 KiB ring of levels 7 / 8 (5 / 4 waves), 6144 with the 32 KiB ring of level 9 (2 waves) */
#ifndef HD_L6_BUCKETS
#define HD_L6_BUCKETS 2560u            /* (1536: ten parse waves per CU instead of eight -- measured below) */
#endif
#define HD_BUCKETS(win_bits, hash_bits) ((win_bits) == 15 ? 6144u : (win_bits) == 14 && (hash_bits) == 13 ? 4096u : (win_bits) == 13 ? HD_L6_BUCKETS : 2560u)

/* Entries of the hash table.  LDS is granted in 1280-byte units, so the table sizes are what fills the units the ring
 * leaves: 1536 entries with the 4 KiB ring of levels 1..2 (7 units, 18 waves per CU instead of 16 with 2048) and the
 * 8 KiB ring of levels 3..4 (12 parse waves instead of 11), 2560 with the 8 KiB ring of levels 5..6 (10 parse waves
 * instead of 9 with 4096: level 6 106 -> 116 GB/s on the FASTQ-like set at the same ratio), 6144 at level 9 (5 waves
 * instead of 4 with 8192).  hash_bits only names the geometry (11: 1536, 12: 2560 / 4096, 13: 6144). */
#define HD_TABLE_34(win_bits, hash_bits)      (((win_bits) <= 13 && (hash_bits) == 11) || ((win_bits) == 14 && (hash_bits) == 13))
#define HD_TABLE_58(win_bits, hash_bits)      ((win_bits) == 13 && (hash_bits) == 12)
#define HD_TABLE_ENTRIES(win_bits, hash_bits) (HD_TABLE_34(win_bits, hash_bits) ? (3u << ((hash_bits) - 2)) : HD_TABLE_58(win_bits, hash_bits) ? 2560u : (1u << (hash_bits)))

/* WORKGROUP LEVELS (3..9, every form): ONE
 * WORKGROUP of HD_WG_WAVES wavefronts per block shares one window and one table in LDS (hd_deflate_wg.hpp) -- the role of
 * hc_matchfinder (lib/libdeflate/hc_matchfinder.h:183-338) and of deflate_compress_lazy_generic (deflate_compress.c:2606-2809)
 * with what a whole CU's LDS holds instead of one wavefront's share of it:
 *   window   DEFLATE's 32 KiB, in a 64 KiB ring (positions are ring offsets; a table entry is p mod 2^16 -- no entry is
 *            "empty": a zeroed or stale one names some position 1..32768 bytes back or is out of range, and the bytes there decide);
 *   table    HD_WG_BUCKETS(level) buckets of HD_WG_WAYS(level) positions, newest first, six-byte key (HD_HASH_SLOT6).  A step's lanes read
 *            their buckets as the steps before left them; of the lanes of a step that share a bucket the highest stores
 *            { itself, the ways - 1 newest before the step };
 *   verify   every candidate -- the byte before the position, if the lane has one in its step (runs), then those of the
 *            bucket -- over HD_WG_VCAP bytes; the longest wins, the nearer on a tie; a match of HD_WG_VCAP bytes is extended to
 *            its full length (<= 258); minimum length HD_WG_MIN_LEN; NO MATCH CROSSES A MULTIPLE OF HD_WG_CUT (a "piece"):
 *            the parse of a piece depends on nothing but the table, so the pieces of a block are parsed side by side, one
 *            wavefront each, and only the table accesses take turns (cut = 1024 costs 0.2 % of the output, tools/mf_explore.c);
 *   lazy     (HD_WG_LAZY(level)) libdeflate's rule (deflate_compress.c:2723-2726) on the lane to the right, with the lengths capped at HD_WG_VCAP:
 *            a match steps aside when 4 (len' - len) + log2(dist) - log2(dist') > 2 for its neighbour's match;
 *   blocks   a DEFLATE block ends at a piece boundary once it holds HD_DYN_BLOCK_TOKENS tokens, or when the token mix has
 *            shifted: libdeflate's observation test (deflate_compress.c:2141-2218) over three classes -- literal, match
 *            below 9 bytes, longer match --, checked every HD_WG_SPLIT_OBS tokens, blocks of at least HD_WG_SPLIT_MIN bytes;
 *   long blocks (MiGz members) are ONE stream with the window sliding through them: no segments.
 * tools/mf_explore.c is the CPU model this geometry was chosen with. */
/* The ladder of the workgroup levels is a ladder of WAYS -- candidates verified per position, the kernel's unit of work --
 * over the same 64 KiB of table (ways x buckets x 2 bytes):
 *   level 3      1 way  x 32768 buckets, greedy      (libdeflate-2's ratio class on every set; libdeflate-1 beaten)
 *   level 4      1 way  x 32768 buckets, lazy
 *   level 5      2 ways x 16384 buckets, lazy
 *   levels 6..9  4 ways x  8192 buckets, lazy        (1.029 / 1.012 / 1.023 of libdeflate-6)
 * Rounds 1-3 ran levels 3..5 in one wavefront's share of LDS (4..8 KiB windows, 1536..2560 entries): 1.12 .. 1.19 of
 * libdeflate-6 on text. */
#define HD_WG_LEVEL        3
#define HD_WG_WAYS(level)    ((level) >= 6 ? 4u : (level) == 5 ? 2u : 1u)
#define HD_WG_BUCKETS(level) (32768u / HD_WG_WAYS(level))
#define HD_WG_LAZY(level)    ((level) >= 4)
#define HD_WG_MAX_WAYS     4
#define HD_WG_WINDOW       32768u
#define HD_WG_RING         65536u
#define HD_WG_VCAP         16u
#define HD_WG_MIN_LEN      5u
#define HD_WG_SPLIT_OBS    512u
#define HD_WG_SPLIT_MIN    5000u
#define HD_WG_CUT          1024u
#define HD_WG_WAVES        16

/* Besides the hash table, which only knows earlier steps, a lane takes the lane just before it as its
 * candidate when that one holds the same four bytes (a run of five equal bytes): the latest occurrence,
 * and on DNA-like data, whose quality strings are full of runs, worth 9 % of the output (0.454 -> 0.415
 * at level 1, 0.311 -> 0.288 with dynamic codes) for three instructions per step.  Looking further back
 * (distances 2..4) adds almost nothing: 0.2875 -> 0.2873.  Levels >= 2; level 1 is the speed level and goes
 * without (it would be 0.411 instead of 0.452 at 211 instead of 221 GB/s). */
#define HD_INTRA_DIST      1

/* a DEFLATE block of the dynamic path is closed at the first step boundary at
 * which it holds at least this many tokens (scratch slab = this + 64 tokens) */
#define HD_DYN_BLOCK_TOKENS (1u << 15)
/* Levels >= 1: a block longer than HD_SEG_LIMIT (a MiGz member, a big zlibutil buffer) is coded as
 * independent HD_SEG_BYTES segments -- each with its own window and codes, each ending in a full flush,
 * an empty final block (03 00) behind the last -- so that one member is work for many wavefronts.
 * The windows are 4..16 KiB, so a fresh window every 0xff00 bytes costs < 1 % of ratio.  Such a block needs
 * room for the worst case of every segment (stored + flush), HD_SEG_WORST(n). */
#define HD_SEG_BYTES       0xff00u     /* not a power of two: waves that start together must not meet on one HBM channel */
#define HD_SEG_LIMIT       (320u << 10)
#define HD_TOKEN_MATCH     0x80000000u /* token: literal byte | MATCH | (len-3)<<16 | (dist-1) */
#define HD_TOKEN_MATCH_TAG 0x81000000u /* what a match token is built with: bit 24 makes bits 16..24 of the word
                                        * 256 + (len - 3), the index of the length in the static code table */
#define HD_LITLEN_MAXBITS  15
#define HD_OFFSET_MAXBITS  15
#define HD_PRECODE_MAXBITS 7

/* worst-case bits one parse step can emit with the static code:
 * 64 tokens x (8+5 + 5+13 = 31 bits) */
#define HD_STEP_MAX_BITS   (64 * 31)

/* stored-block framing cost: 5 bytes per <=65535-byte block, at least one */
#define HD_STORED_SIZE(n)  ((n) + 5u * ((n) == 0 ? 1u : (((n) + 65534u) / 65535u)))
/* worst-case payload of a block of n bytes coded as `seg`-byte segments: every segment stored (5 bytes per <= 65535
 * bytes) + its flush suffix; `flush` = the member itself is in flush form (no 03 00 behind) */
#define HD_SEGN_COUNT(n, seg)  (((n) + (seg) - 1u) / (seg))
#define HD_SEGN_WORST(n, seg, flush) \
	(((n) / (seg)) * (HD_STORED_SIZE(seg) + 5u) + \
	 ((n) % (seg) ? HD_STORED_SIZE((n) % (seg)) + 5u : 0u) + ((flush) ? 0u : 2u))
#define HD_SEG_COUNT(n)        HD_SEGN_COUNT(n, HD_SEG_BYTES)
#define HD_SEG_WORST(n, flush) HD_SEGN_WORST(n, HD_SEG_BYTES, flush)
/* LATENCY MODE (HD_FRAME_LATENCY in the frame argument; the LD_PRELOAD hook and the per-block codecs use it): a batch
 * too small to fill 4608 resident wavefronts -- htslib's 4..16 worker threads hand the hook one 0xff00-byte block each
 * -- is coded with SEVERAL wavefronts per block: every block longer than HD_LAT_SEG_BYTES(level) becomes independent
 * segments of that size, coded, flushed and stitched exactly as the long blocks above.  A 0xff00-byte block is 16
 * segments at level 1 (4080 bytes: the 4 KiB window never held more) and 8 at the dynamic levels; its worst case,
 * 16 x 4090 + 2 + 26, still fits a BGZF member. */
#define HD_LAT_SEG_BYTES(level) ((level) <= 1 ? 4080u : 8160u)
/* ... and at the dynamic levels a latency segment is PARSED in parts of HD_LAT_PART_BYTES, one wavefront each (its own table
 * and window, no match out of the part's own wavefront's sight), while ONE wavefront builds one code over the tokens of all
 * parts and emits them as one DEFLATE block.  The parse of a wavefront that is alone on its SIMD is serial work at ~0.6 us
 * per 64-byte step -- 105 of the 190 us a level-2 batch of sixteen 0xff00-byte blocks took.  A second Huffman header per 4080
 * bytes -- segments half the size, the alternative -- costs 4 % of ratio; parts cost under 1 %.
 *   A part behind the first of its segment is PRIMED: its wavefront starts HD_LAT_PRIME_BYTES early and runs those steps
 * without making tokens (positions enter the table, the ring fills), so the first matches may reach back across the border.
 * The 16-block test set of tools (FASTQ-like | text), level 2 and 6, bytes out / bytes in:
 *     whole 8160-byte segments                 0.2979 | 0.432    level 6 0.2820
 *     2 parts of 4080, not primed              0.3064 | 0.4396
 *     2 parts of 4080, primed with 512         0.3007 | 0.4347           0.2841
 *     4 parts of 2048, primed with 512 (this)  0.3009 | 0.4410           0.2832
 *     8 parts of 1024, primed with 512         0.3002 | 0.4473           0.2828  (146 instead of 155 us per batch; not taken)
 * A SEGMENT that is not the first of its block is primed the same way, at every level (at level 1 it is the only priming
 * there is): it starts 512 bytes inside its predecessor.  DEFLATE allows the reach -- the segments of a member are one
 * stream --, and a fresh window every 4080 bytes was what latency mode cost: FASTQ-like level 1 0.4812 -> 0.4586, level 2
 * 0.3009 -> 0.2954, level 6 0.2832 -> 0.2800; text level 1 0.6537 -> 0.6313.
 * Parts and segments of less than 64 bytes go without priming (the CRC lanes want 16 bytes of their own).  HD_LAT_PRIME:
 * o = the bytes of the block ahead of the part / segment, n = its own bytes.  HD_LAT_PARTS: 0 = not a latency segment of a
 * dynamic level, parsed whole. */
#define HD_LAT_PART_BYTES  2048u
#define HD_LAT_PRIME_BYTES 512u
#define HD_LAT_PRIME(o, n) ((o) >= HD_LAT_PRIME_BYTES && (n) >= 64u ? HD_LAT_PRIME_BYTES : 0u)
#ifndef HD_LAT_SEG_PRIME
#define HD_LAT_SEG_PRIME   1           /* 0: segments are not primed, only parts (A/B builds: tools/exp_seg_prime.sh) */
#endif
#define HD_LAT_PARTS_MAX   4u
#define HD_LAT_PARTS(level, seg) ((level) >= 2 && (level) < HD_WG_LEVEL && (seg) == HD_LAT_SEG_BYTES(level) ? HD_LAT_PARTS_MAX : 0u)

/* one compressed stream handed to the inflate kernel must be shorter than this: it keeps stream positions
 * as 32-bit bit counts (8 n + 64 + 24 < 2^32) */
#define HD_INFLATE_MAX_IN  (1u << 28)

/* result codes of the inflate path = enum libdeflate_result
 * (lib/libdeflate/libdeflate.h:193-208), which libdeflate_inflate
 * (lib/zlibutil.c:194-204) hands straight back to the applet */
#define HD_OK                  0
#define HD_BAD_DATA            1
#define HD_SHORT_OUTPUT        2
#define HD_INSUFFICIENT_SPACE  3

#endif
