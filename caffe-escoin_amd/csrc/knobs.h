// knobs.h -- the library's environment switches, by build flavour.
//
// The PRODUCT build (csrc/Makefile, what __graft_entry__.build() makes) reads two environment variables and
// neither can change a result: ESCOIN_VERBOSE (diagnostics on stderr) and TMPDIR (where the code object manager
// may put temporaries, jit_module.cpp).  Everything else is a compile-time constant there: ESC_KNOB("NAME", d)
// is the literal d, and the name does not even reach the object file (tests/test_capi_cpu.py greps for it).
//
//   -DESCOIN_EXPERIMENTS  (tools/mkabl.sh exp -> libescoin_exp.so): the tuning switches are live -- tilings, buffer
//                         counts, kernel selection, DMA spread ...  Results stay within the parity tolerance but
//                         may differ in the last bits (another kernel or summation order); for sweeps and A/B runs.
//   -DESCOIN_ABLATIONS    (tools/mkabl.sh -> libescoin_abl.so): implies the above, plus the in-kernel stamp profile
//                         (ESCOIN_PROF) and the switches that make results WRONG on purpose to time what is left
//                         (ESCOIN_DBG, ESCOIN_JIT_ABL, ESCOIN_DENSE_ABL).  Never a product.
#ifndef ESCOIN_KNOBS_H_
#define ESCOIN_KNOBS_H_

#include <cstdlib>

#if defined(ESCOIN_ABLATIONS) && !defined(ESCOIN_EXPERIMENTS)
#define ESCOIN_EXPERIMENTS 1
#endif

#ifdef ESCOIN_EXPERIMENTS
namespace escoin {
inline long knob_long(const char *name, long dflt) {
  const char *e = getenv(name);
  return e ? atol(e) : dflt;
}
inline double knob_double(const char *name, double dflt) {
  const char *e = getenv(name);
  return e ? atof(e) : dflt;
}
}  // namespace escoin
#define ESC_KNOB(name, dflt) (::escoin::knob_long("ESCOIN_" name, (dflt)))
#define ESC_KNOB_F(name, dflt) (::escoin::knob_double("ESCOIN_" name, (dflt)))
#define ESC_KNOB_SET(name) (getenv("ESCOIN_" name) != nullptr)
#else
#define ESC_KNOB(name, dflt) ((long)(dflt))
#define ESC_KNOB_F(name, dflt) ((double)(dflt))
#define ESC_KNOB_SET(name) (false)
#endif

// wrong-result switches: ablation builds only
#ifdef ESCOIN_ABLATIONS
#define ESC_ABL_KNOB(name) ((int)::escoin::knob_long("ESCOIN_" name, 0))
#define ESC_DBG(a, bits) ((a).dbg & (bits))
#else
#define ESC_ABL_KNOB(name) (0)
#define ESC_DBG(a, bits) (0)
#endif

#endif  // ESCOIN_KNOBS_H_
