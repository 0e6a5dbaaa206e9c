// Internal helpers shared by the translation units of libglsdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
#include <string>

#include "../../include/glsdet_hip.h"

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace glsdet {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define GLS_FAIL(code, ...)          \
  do {                               \
    ::glsdet::set_error(__VA_ARGS__); \
    return (code);                   \
  } while (0)
#define GLS_HIP(expr)                                                             \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) GLS_FAIL(GLSDET_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

inline int dtype_size(int dt) { return dt == GLSDET_F16 ? 2 : 4; }

// Host-side check that a view is well formed and that every byte it can address lies in
// its allocation.  `what` names the operand in the error text.
int check_view(const glsdet_view& v, const char* what, bool need16 = true);
// same extent (n,h,w,c)?
inline bool same_extent(const glsdet_view& a, const glsdet_view& b) {
  return a.n == b.n && a.h == b.h && a.w == b.w && a.c == b.c;
}

// ---- plan recording -------------------------------------------------------------------
struct OpRecord {
  int kind;                 // 0 conv, 1 focus, 2 maxpool, 3 resample, 4 nonlocal, 5 decode, 6 nms
  int branch = 0;           // 0 = main sequence; ops of different non-zero branches between two
                            // main ops are independent and may overlap (set by submit())
  double flops, bytes;      // algorithmic 2*MACs and min HBM bytes of this op
  std::string name;         // kernel family / tile name
  std::function<int(hipStream_t)> launch;
};
// If the calling thread is recording a plan the op is appended and 0 returned, otherwise it
// is launched on `stream` right away.
int submit(OpRecord&& op, void* stream);

}  // namespace glsdet
