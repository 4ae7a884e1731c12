/* oracle/ref_plan.h -- TEST INFRASTRUCTURE ONLY: the aligned state shared by the two translation
 * units of oracle/_ref/libescoin_ref.so (ref_driver.cpp: the plain loop nest; ref_blocked.cpp: the
 * register-blocked kernel).  Holds what the reference's WeightAlign leaves in the layer
 * (base_conv_layer.cpp:46-273), so that a timed forward does not redo dense -> CSR. */
#ifndef ESCOIN_ORACLE_REF_PLAN_H_
#define ESCOIN_ORACLE_REF_PLAN_H_

#include <vector>

#include "sconv_oracle.h"

typedef void (*unit_stride_fn)(const float *, const int **, const int **, const float **, int,
                               const float *, float *, int, int, float *, int, int);

struct ref_plan {
  oracle_conv_geom g;
  int OH, OW, Cg, Mg, kdim;
  long plen;
  /* plain path: per-group CSR with stretched indices (WeightAlign's non-blocked branch) */
  std::vector<int> rowptr, colidx;
  std::vector<float> values;
  /* blocked path (BLOCKED_SCONV branch) */
  unit_stride_fn kernel;
  int ncolblocks;        /* over all groups */
  std::vector<std::vector<int> > b_rowptr, b_colidx;
  std::vector<std::vector<float> > b_values;
  std::vector<const int *> b_rowptr_p, b_colidx_p;
  std::vector<const float *> b_values_p;
};

#endif
