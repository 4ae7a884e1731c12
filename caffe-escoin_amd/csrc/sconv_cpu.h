// sconv_cpu.h -- the host (Caffe::CPU mode) half of the library: internal interface between the
// dispatcher (sconv_cpu.cpp) and the two ISA flavours of the kernel translation unit
// (sconv_cpu_kernel.cpp compiled with -mavx2 -mfma and with -mavx512f).  Not part of the C ABI.
//
// The path it serves:  ConvolutionLayer<Dtype>::Forward_cpu           conv_layer.cpp:25-63
//                        -> BaseConvolutionLayer::forward_cpu_sconv    base_conv_layer.cpp:569-661
//                           -> caffe_cpu_sconv<Dtype>                  math_functions.cpp:128-176
//                        -> forward_cpu_bias                           base_conv_layer.cpp:663-669
// Nothing here includes, links or loads anything under oracle/.
#ifndef ESCOIN_SCONV_CPU_H_
#define ESCOIN_SCONV_CPU_H_

#include <cstddef>

namespace escoin {
namespace cpu {

// One conv group of one image on the reference's shared-halo padded layout
// (C x (H + pad_h) x (W + pad_w) floats + tail, base_conv_layer.cpp:71,596-620).
template <typename T>
struct GroupJob {
  const T *in;          // this group's first channel in the padded image
  const int *rowptr;    // Mg + 1, offsets into off / val
  const int *off;       // per nonzero: (ic * PH + kr * dil_h) * PW + kc * dil_w  (dilation folded in)
  const T *val;         // per nonzero, CSR order (ascending columns: the reference's summation order)
  const T *bias;        // this group's Mg biases or nullptr; added ONCE after the sum (conv_layer.cpp:55-58)
  T *out;               // this group's first output plane, Mg x OH x OW
  int m_begin, m_end;   // output channels of the group this call covers
  int OH, OW, PW;       // PW = W + pad_w: pitch of a padded row
  int stride_h, stride_w;
  int relu;             // ConvolutionReLU: max(x + bias, 0)
  int exact_reads;      // 1: never read past the last input element an output needs (drop-in callers hand over
                        // buffers of exactly the reference's length); 0: a vector of slack follows the image
  T *scratch;           // >= scratch_len<T>(OH, PW, OW) elements, private to the calling thread
  // Channel blocking (stride 1 only; the reference's register-blocked kernel does the same, sconv.hpp:57-589 "column
  // blocking" with partial sums in scratch): a tile's input window of ALL channels does not fit L1 on the larger
  // layers, so the channels go by in blocks -- block outer, output channel inner, the tile's partial sums of every
  // output channel parked in `partial` between blocks.  A row's nonzeros are in ascending channel order, so block after
  // block continues the same sum in the same order: results stay bit-identical.
  const int *blk_ptr;   // nullptr = no blocking; else [m * (n_blk + 1) + b] = first nonzero of row m in channel block b
  int n_blk;
  T *partial;           // >= (m_end - m_begin) * partial_elems_per_row() elements, private to the calling thread
  // Several images per job (small images only, images_per_job below): image i reads in + i * in_stride and writes
  // out + i * out_stride.  A 7 x 7 image is four vectors -- four independent multiply-adds per nonzero against a
  // latency of four cycles on two pipes -- so two or three images share every broadcast weight and fill the registers.
  int n_img;            // 1 .. images_per_job
  size_t in_stride, out_stride;
};

size_t scratch_elems(int OH, int PW);
constexpr size_t kPartialElemsPerRow = 16 * 16;   // one tile of the widest flavour (14 vectors x 16 floats), rounded up

// Input channels per block for this geometry in the named flavour, 0 = do not block.  span_rows = (KH - 1) * dil_h,
// avg_row_nnz = nonzeros per output channel (a block with a handful of nonzeros per row costs more in parked sums than
// it saves in L1 misses).
// Images a job should carry for this geometry in the named flavour (1 = one at a time).
template <typename T> int images_per_job_avx2(int OH, int OW, int PW, int stride_h, int stride_w);
template <typename T> int images_per_job_avx512(int OH, int OW, int PW, int stride_h, int stride_w);
// Bytes of ONE channel of a tile's input window for one image (whole padded rows the tile's pixels span + the kernel's).
template <typename T> long window_bytes_avx2(int OH, int OW, int PW, int span_rows);
template <typename T> long window_bytes_avx512(int OH, int OW, int PW, int span_rows);
template <typename T> int channel_block_avx2(int OH, int OW, int PW, int span_rows, int Cg, double avg_row_nnz);
template <typename T> int channel_block_avx512(int OH, int OW, int PW, int span_rows, int Cg, double avg_row_nnz);

// Runs one job.  Each output is sum = fma(val[j], in[...], sum) over the row's nonzeros in CSR order starting from
// zero, then + bias, then ReLU -- lane for lane the arithmetic of caffe_cpu_sconv, so results are bit-identical to it
// (and to oracle/, which tests/ checks; this file does not know the oracle exists).
template <typename T> void run_group_avx2(const GroupJob<T> &job);
template <typename T> void run_group_avx512(const GroupJob<T> &job);

}  // namespace cpu
}  // namespace escoin
#endif
