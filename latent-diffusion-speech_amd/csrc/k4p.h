// K4P: the k-interleaved, padded activation layout used between the UNet's kernels.
//
//   tensor [B][C][T] (C % 8 == 0)  ->  [B][C/8][2][T+2][4]
//   channel c = 8q + 2j + h, frame t  ->  ((((b*(C/8) + q)*2 + h)*(T+2) + (t+1))*4 + j
//
// * v_mfma_f32_32x32x2_f32 wants, per lane, ONE scalar of A/B whose k index is 2*kp + (lane>>5).  With this
//   layout the 4 floats at one (row, frame) are the operands of four consecutive MFMAs for lane half h, so an
//   LDS copy of a row segment serves ds_read_b128 operand loads with no transpose and no VALU.
// * frames -1 and T of every row are zero: they are the k3 convolutions' padding, so activation tiles are pure
//   linear copies (LDS-DMA).  Every kernel that writes a K4P tensor also writes its two pad frames.
// * channel concatenation = concatenation of 8-channel blocks, so the UNet's skip-concat is a second base pointer.
#pragma once
#include <hip/hip_runtime.h>

namespace lds {

__host__ __device__ inline long long k4p_index(int C, int T, int b, int c, int t) {
    return ((((long long)b * (C >> 3) + (c >> 3)) * 2 + (c & 1)) * (T + 2) + (t + 1)) * 4 + ((c & 7) >> 1);
}
__host__ __device__ inline long long k4p_floats(int B, int C, int T) { return (long long)B * C * (T + 2); }

// 8-byte write-through store (global_store_dwordx2 ... sc1): the line goes to memory while the kernel is still computing
// instead of staying dirty in the XCD's L2 until the end-of-kernel write-back, which the next launch has to wait for
// (MI355X_MICROARCH.md price list, row "boundary": + bytes / 6 TB/s when the predecessor leaves dirty lines).  Every 64-byte
// line of a K4P row is written whole by one wave instruction (lanes = consecutive frames, both lane halves fill an entry).
typedef float k4p_f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void k4p_store_wt(float* p, k4p_f32x2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Ragged batches: lens[b] = valid frames of utterance b at the UNet's input resolution (device int32 [B]; nullptr = every utterance has the
// buffer's length T).  Level l of the UNet halves a length l times the way its stride-2 convolutions do.  Frames at and beyond an
// utterance's length hold zeros in every activation tensor -- they are the convolutions' zero padding, exactly as when the utterance
// runs alone -- so every kernel that writes a tensor writes zeros there, and GroupNorm statistics / attention keys stop at the length.
static __device__ __forceinline__ int ragged_len(const int* lens, int b, int lvl, int T) {
    if (!lens) return T;
    int n = lens[b];
    for (int i = 0; i < lvl; ++i) n = (n - 1) / 2 + 1;
    return n < T ? n : T;
}

}  // namespace lds
