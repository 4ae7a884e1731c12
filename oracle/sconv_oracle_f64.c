/*
 * oracle/sconv_oracle_f64.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The Dtype = double instantiation of the oracle: the reference instantiates caffe_cpu_sconv for double as well
 * (math_functions.cpp:194-199) from the same template text as for float, so the double oracle is made the same way --
 * it IS sconv_oracle.c, compiled once more with `float` read as `double`, `fmaf` as `fma` and every function name
 * suffixed _f64.  No line of arithmetic exists twice, so whatever pins the float restatement (oracle/_ref, the golden
 * fixtures) pins this text too; tests/test_oracle.py additionally checks the two instantiations against each other on
 * integer-valued data, where both are exact.  (sconv.hpp, the header oracle/_ref compiles, is float-only: there is no
 * reference double kernel to build here.)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define float double
#define fmaf fma
#define ESCOIN_SCONV_ORACLE_H_F64_
#define oracle_conv_geom oracle_conv_geom_f64
#define oracle_out_dim oracle_out_dim_f64
#define oracle_padded_len oracle_padded_len_f64
#define oracle_dense2csr oracle_dense2csr_f64
#define oracle_stretch oracle_stretch_f64
#define oracle_pad_input oracle_pad_input_f64
#define oracle_sconv oracle_sconv_f64
#define oracle_bias oracle_bias_f64
#define oracle_dense_gemm oracle_dense_gemm_f64
#define oracle_conv_forward oracle_conv_forward_f64
#define oracle_conv_forward_nogate oracle_conv_forward_nogate_f64

#include "sconv_oracle.c"
