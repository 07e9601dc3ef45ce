// glu_math.hpp -- the activation of the gated MLP, shared by the element-wise kernels (decoder_ops.hip) and the GLU epilogue of the
// gate|up GEMM (gemm.hip) so that the fused and the separate path produce the same bits.
//   SiLU       act(g) = g * sigmoid(g)                      (LlamaMLP, modeling_llama.py:227-258)
//   tanh-GELU  act(g) = 0.5 g (1 + tanh(u)) = g * sigmoid(2u),  u = sqrt(2/pi) (g + 0.044715 g^3)   (GemmaMLP, gelu_pytorch_tanh)
// sigmoid(x) = 1 / (1 + 2^(-x log2 e)) as one v_exp_f32 and one v_rcp_f32 (1 ulp each; the results are rounded to bf16).  libm's tanhf
// in the GEMM epilogue made the fused Gemma projection slower than the two separate launches.
#pragma once
#include <hip/hip_runtime.h>

namespace ecgb {

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

template <bool GELU_TANH>
__device__ __forceinline__ float glu_act(float g)
{
    if (GELU_TANH) { const float k = 0.7978845608028654f; return g * fast_sigmoid(2.f * k * (g + 0.044715f * g * g * g)); }
    return g * fast_sigmoid(g);
}

template <bool GELU_TANH>
__device__ __forceinline__ float glu_act_grad(float g)
{
    if (GELU_TANH) {
        const float k = 0.7978845608028654f;
        const float t = 2.f * fast_sigmoid(2.f * k * (g + 0.044715f * g * g * g)) - 1.f;      // tanh(u)
        return 0.5f * (1.f + t) + 0.5f * g * (1.f - t * t) * k * (1.f + 3.f * 0.044715f * g * g);
    }
    const float s = fast_sigmoid(g);
    return s * (1.f + g * (1.f - s));
}

}  // namespace ecgb
