// hd_deflate_dynamic.hpp -- levels >= 2: dynamic Huffman (and lazy parse from 5).
// PLACEHOLDER WIRING for the first GPU bring-up: until the dynamic kernel lands,
// levels >= 2 run the level-1 kernel (the CPU twin does the same), so the
// output is valid and twin-identical at every level.
#pragma once
#include "hd_deflate_static.hpp"

namespace hd {

inline uint64_t dynamic_scratch_bytes(uint32_t, uint32_t, int) { return 0; }

inline int launch_deflate_dynamic(const DeflateArgs &a, int level, hipStream_t st)
{
	DeflateArgs b = a;
	b.level = level;
	hipLaunchKernelGGL((k_deflate_static<HD_L1_WIN_BITS, HD_L1_HASH_BITS>), dim3(a.nblocks), dim3(64), 0, st, b);
	return 0;
}

} // namespace hd
