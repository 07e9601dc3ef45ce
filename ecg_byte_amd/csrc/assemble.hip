// assemble.hip -- ECG token ids -> LLM input rows on the device.
//
// Reproduces ECGTokenDataset.__getitem__/_prepare_training/_prepare_inference of the reference
// (ecg_byte/data_loader.py:80,91-132) for a whole batch in one launch:
//   sig      = LUT[encode(...)]                                   (data_loader.py:80)
//   space    = pad_to_max - len(Q) - len(A)                       (:103-104)
//   row      = [pad]*(space-n) + [bos, <sig_start>] + sig[:space] + [<sig_end>] + Q + A + [eos]
//   labels   = [-100]*(everything before A) + A + [eos]           (:115)
//   mask     = row != pad_id  (by VALUE, as create_attention_like_mask does, :21-22)
//   position = cumsum(mask) - 1, 0 where mask == 0                (:25-31)
// Row length is pad_to_max + 4 (asserted at :123).  One workgroup per row; the by-value mask
// makes position_ids a genuine prefix sum, done with a wave scan + LDS carry.
#include <hip/hip_runtime.h>

#include <string>

#include "tokenizer.hpp"

namespace {

struct AssembleArgs {
    const uint32_t *ids;        // batch x ids_stride encoder output
    size_t ids_stride;
    const uint32_t *counts;     // batch
    const int32_t *lut;         // tokenizer id -> LLM id
    uint32_t lut_len;
    const int32_t *q_ids;       // concatenated question ids
    const uint32_t *q_off;      // batch + 1
    const int32_t *a_ids;       // concatenated answer ids (training only)
    const uint32_t *a_off;      // batch + 1
    int32_t pad_id, bos_id, eos_id, sig_start_id, sig_end_id;
    uint32_t pad_to_max;
    uint32_t row_len;           // training: pad_to_max + 4; inference: out stride
    int inference;
    int64_t *input_ids;         // batch x row_len
    float *attn_mask;           // batch x row_len
    int64_t *labels;            // batch x row_len (training)
    int64_t *position_ids;      // batch x row_len (training)
    uint32_t *lengths;          // batch (inference: valid length of each row)
};

__global__ __launch_bounds__(256) void assemble_kernel(AssembleArgs A)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    const uint32_t b = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t q0 = A.q_off[b], nq = A.q_off[b + 1] - q0;
    const uint32_t a0 = A.inference ? 0u : A.a_off[b];
    const uint32_t na = A.inference ? 0u : A.a_off[b + 1] - a0;
    const uint32_t count = A.counts[b];
    const uint32_t avail = (uint32_t)min((size_t)count, A.ids_stride);
    uint32_t n_used, npad, len;
    if (A.inference) {   // [bos, <sig_start>] + sig + [<sig_end>] + Q   (data_loader.py:92)
        n_used = avail; npad = 0; len = 2 + n_used + 1 + nq;
        if (len > A.row_len) { n_used -= min(n_used, len - A.row_len); len = 2 + n_used + 1 + nq; }
        if (tid == 0) A.lengths[b] = len;
    } else {
        const uint32_t space = A.pad_to_max - nq - na;   // host guarantees nq + na <= pad_to_max
        n_used = min(avail, space);
        npad = space - n_used;
        len = A.row_len;
    }
    const uint32_t sig_begin = npad + 2, sig_end_pos = sig_begin + n_used;
    const uint32_t q_begin = sig_end_pos + 1, a_begin = q_begin + nq, eos_pos = a_begin + na;
    const uint32_t *ids = A.ids + (size_t)b * A.ids_stride;
    const size_t row = (size_t)b * A.row_len;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < A.row_len; base += 256) {
        const uint32_t t = base + tid;
        int32_t tokv = A.pad_id;
        if (t < len) {
            if (t < npad) tokv = A.pad_id;
            else if (t == npad) tokv = A.bos_id;
            else if (t == npad + 1) tokv = A.sig_start_id;
            else if (t < sig_end_pos) { uint32_t k = ids[t - sig_begin]; tokv = (k < A.lut_len) ? A.lut[k] : A.pad_id; }
            else if (t == sig_end_pos) tokv = A.sig_end_id;
            else if (t < a_begin) tokv = A.q_ids[q0 + (t - q_begin)];
            else if (t < eos_pos) tokv = A.a_ids[a0 + (t - a_begin)];
            else tokv = A.eos_id;
        }
        const uint32_t m = (t < len && tokv != A.pad_id) ? 1u : 0u;
        uint32_t incl = m;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t v = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += v;
        }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t k = 0; k < wv; ++k) before += s_wave[k];
        if (t < A.row_len) {
            A.input_ids[row + t] = (t < len) ? (int64_t)tokv : (int64_t)A.pad_id;
            A.attn_mask[row + t] = m ? 1.0f : 0.0f;
            if (!A.inference) {
                A.labels[row + t] = (t < a_begin) ? (int64_t)-100 : (int64_t)tokv;
                A.position_ids[row + t] = m ? (int64_t)(before + incl) - 1 : (int64_t)0;
            }
        }
        __syncthreads();
        if (tid == 0) s_carry += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}

}  // namespace

extern "C" int ecgb_assemble_hip(const uint32_t *ids_dev, size_t ids_stride, const uint32_t *counts_dev,
                                 size_t batch, const int32_t *lut_dev, size_t lut_len,
                                 const int32_t *q_ids_dev, const uint32_t *q_offsets_dev,
                                 const int32_t *a_ids_dev, const uint32_t *a_offsets_dev,
                                 int32_t pad_id, int32_t bos_id, int32_t eos_id, int32_t sig_start_id,
                                 int32_t sig_end_id, uint32_t pad_to_max, int inference,
                                 uint32_t row_len, int64_t *input_ids_dev, float *attn_mask_dev,
                                 int64_t *labels_dev, int64_t *position_ids_dev, uint32_t *lengths_dev,
                                 void *stream)
{
    if (batch == 0) return ECGB_OK;
    if (!ids_dev || !counts_dev || !lut_dev || !q_offsets_dev || !input_ids_dev || !attn_mask_dev ||
        (!inference && (!a_offsets_dev || !labels_dev || !position_ids_dev)) || (inference && !lengths_dev)) {
        ecgb::set_error("ecgb_assemble_hip: NULL argument");
        return ECGB_ERR_INVALID;
    }
    if (!inference && row_len != pad_to_max + 4) {
        ecgb::set_error("ecgb_assemble_hip: training rows are pad_to_max + 4 long (data_loader.py:123)");
        return ECGB_ERR_INVALID;
    }
    AssembleArgs A;
    A.ids = ids_dev; A.ids_stride = ids_stride; A.counts = counts_dev;
    A.lut = lut_dev; A.lut_len = (uint32_t)lut_len;
    A.q_ids = q_ids_dev; A.q_off = q_offsets_dev; A.a_ids = a_ids_dev; A.a_off = a_offsets_dev;
    A.pad_id = pad_id; A.bos_id = bos_id; A.eos_id = eos_id;
    A.sig_start_id = sig_start_id; A.sig_end_id = sig_end_id;
    A.pad_to_max = pad_to_max; A.row_len = row_len; A.inference = inference;
    A.input_ids = input_ids_dev; A.attn_mask = attn_mask_dev; A.labels = labels_dev;
    A.position_ids = position_ids_dev; A.lengths = lengths_dev;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ecgb::set_error(std::string("assemble_kernel launch: ") + hipGetErrorString(e));
        return ECGB_ERR_HIP;
    }
    return ECGB_OK;
}
