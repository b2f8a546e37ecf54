// minicom_amd/csrc/cindex.hpp -- geometry of the Stage-2 contig index shared by its build (cindex.hip) and its lookups
// (realign.hip).  See cindex.hip for the structure.
#pragma once
#include "mcom_dev.hpp"

#define MAXDICT 16

__host__ __device__ static inline int dict_layout(int L, int ininumdict, int *start, int *len)
{
	// setglobalarrays_realign (kthread_hash_realign.c:153-206)
	const int len_t = L <= 80 ? 11 : 17;
	int nd = L / len_t;
	if (ininumdict > 1 && ininumdict < nd) nd = ininumdict;
	if (nd > MAXDICT) nd = MAXDICT;
	start[0] = (ininumdict > 0 && ininumdict < nd) ? L / 2 - (len_t * nd) / 2 : 0;
	len[0] = len_t;
	for (int i = 1; i < nd; ++i) { start[i] = start[i - 1] + len_t; len[i] = len_t; }
	return nd;
}

#define CIX_WAYS 7u                         // entries per line of eight words
// word 0 of a line: bits 0-7 entries in the line; CIX_MORE: entries were pushed past it (the lookup goes on to the next line);
// CIX_HEAVY: the keys whose home this line is have so many entries (a repeat) that they live in a run of lines of the
// extension area instead: bits 10-31 = lines of the run, bits 32-63 = its first line (counted from the extension's start)
#define CIX_MORE 0x100ull
#define CIX_HEAVY 0x200ull
// an entry = tag (12 bits) | contig index | position in the contig.  The 52 bits below the tag are shared: up to 2^24 contigs
// the position has 28 bits (the layout of rounds 1 and 2); a larger set (a 500 M-read job) takes the bits its contig indices
// need from the position field.  Both sides derive the cut from the number of contigs of the set; the build refuses a
// position that does not fit (header word 1).
#define CIX_TAG_SHIFT 52
__host__ __device__ static inline int cix_pbits(uint32_t n_contigs) { int cb = 24; while (cb < 32 && (1ull << cb) <= (uint64_t)n_contigs) ++cb; return CIX_TAG_SHIFT - cb; }
#define CIX_MAX_PARTS 65535u
#define CIX_HEAD_WORDS 8                    // d_keys[0] = lines of the extension area in use, d_keys[1] != 0: a contig too long for the position field
// n_lines: lines per partition.  Multi-GPU: the ONE index over all contigs is cut BY KEY into n_owners equal shares, share q
// (n_parts partitions, built and looked up on rank q alone) holds the keys whose hash falls into it; owner = this rank's share
struct CixGeom { uint32_t n_parts, n_lines; int L, nd, klen, maxoff, pbits; uint32_t n_owners, owner; int ds[MAXDICT]; };

static inline int cix_geom(int L, int ininumdict, CixGeom &g)
{
	int len[MAXDICT];
	g.L = L; g.nd = dict_layout(L, ininumdict, g.ds, len);
	if (g.nd < 1) return -1;
	g.klen = len[0];
	int mo = 0;
	for (int l = 0; l < g.nd; ++l) {
		if (g.ds[l] > mo) mo = g.ds[l];
		if (g.ds[l] > 0 && L - g.ds[l] - g.klen > mo) mo = L - g.ds[l] - g.klen;
	}
	g.maxoff = mo;
	g.pbits = 28;
	g.n_owners = 1; g.owner = 0;
	return 0;
}

// d_keys = [ 8 words of header | n_parts * n_lines lines of 8 words | extension lines ];
// geom = n_parts | n_lines << 16 | (n_owners - 1) << 32 | owner << 40
#define CIX_MAX_LINES 12000u                // the line counters of a partition live in LDS
#define CIX_MAX_OWNERS 256u
static_assert(CIX_MAX_PARTS <= 0xFFFFu && CIX_MAX_LINES <= 0xFFFFu && CIX_MAX_OWNERS <= 256u, "the fields of the geometry word");
__host__ __device__ static inline uint64_t cix_pack(uint32_t n_parts, uint32_t n_lines, uint32_t n_owners, uint32_t owner)
{
	return (uint64_t)n_parts | ((uint64_t)n_lines << 16) | ((uint64_t)(n_owners - 1) << 32) | ((uint64_t)owner << 40);
}
static inline void cix_unpack(uint64_t geom, CixGeom &g)
{
	g.n_parts = (uint32_t)(geom & 0xFFFFu); g.n_lines = (uint32_t)((geom >> 16) & 0xFFFFu);
	g.n_owners = (uint32_t)((geom >> 32) & 0xFFu) + 1; g.owner = (uint32_t)((geom >> 40) & 0xFFu);
}

// where a key lives: the share (rank) that owns it, its partition inside that share, the 16 bits that pick its home line inside
// the partition; its 12-bit tag comes from a second multiplier.  The top 32 bits of the hash are a fixed-point fraction: the
// share is its integer part times n_owners, the partition the same of what is left -- no division, and one share (the
// single-GPU index) gives the partition of rounds 1 and 2.
__device__ __forceinline__ void cix_hash(uint64_t key, uint32_t n_owners, uint32_t n_parts, uint32_t &own, uint32_t &part, uint32_t &h16)
{
	const uint64_t h = key * 0x9E3779B97F4A7C15ull;
	const uint64_t f = (h >> 32) * (uint64_t)n_owners;
	own = (uint32_t)(f >> 32);
	part = (uint32_t)(((f & 0xFFFFFFFFull) * (uint64_t)n_parts) >> 32);
	h16 = (uint32_t)(h >> 16) & 0xFFFFu;
}
__device__ __forceinline__ uint64_t cix_tag(uint64_t key) { return (key * 0xD6E8FEB86659FD93ull) >> CIX_TAG_SHIFT; }
__device__ __forceinline__ uint32_t cix_home(uint32_t h16, uint32_t n_lines) { return (h16 * n_lines) >> 16; }
