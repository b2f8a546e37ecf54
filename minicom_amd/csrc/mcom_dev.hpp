// minicom_amd/csrc/mcom_dev.hpp -- shared host/device helpers for libmcom_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include "../../include/mcom.h"

struct mcom_ctx {
	int device;
	hipStream_t stream;
	std::string err;
	// grow-only device workspace
	void *ws; size_t ws_bytes;
	int n_cu;
};

int mcom_fail(mcom_ctx *ctx, int code, const char *fmt, ...);
int mcom_ws_reserve(mcom_ctx *ctx, size_t bytes);

// internal helpers shared between translation units (sort.hip)
int mcom_scan_u32(mcom_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch);
size_t mcom_scan_scratch_elems(size_t n);
size_t mcom_sort_ws_bytes(size_t n);
int mcom_sort_by_x(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n, int bits, void *ws);

#define MCOM_HIP(ctx, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) \
	return mcom_fail(ctx, MCOM_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); } while (0)
#define MCOM_LAUNCH_CHECK(ctx) MCOM_HIP(ctx, hipGetLastError())

#define U64MAX 0xFFFFFFFFFFFFFFFFull

// hash64: invertible mix on 2k bits (reference sketch.c:27-37)
__device__ __forceinline__ uint64_t mcom_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

// 32-bit form of the same mix for 2k <= 32 (mask fits in one register)
__device__ __forceinline__ uint32_t mcom_hash64_lo(uint32_t key, uint32_t mask)
{
	// every shift of the 64-bit form that can move bits across bit 32 is dead once key < 2^32:
	// (key << s) & mask keeps only the low 32 bits, key >> s never imports high bits.
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

static inline int mcom_words_per_read(int L) { return (2 * L + 63) / 64; }
