// Wrapper translation unit: compiles the reference's header-only EESM helper WHERE IT LIES
// (/root/reference/src/utility/eesm-effective-sinr.h, added with -I by oracle/Makefile) and
// exports it with C linkage for the parity tests.  No reference source is copied here.
#include <stdexcept>
#include <vector>

#include "utility/eesm-effective-sinr.h"

extern "C" double ref_eesm_effective_sinr(const double* sinr_db, int n) {
  std::vector<double> v(sinr_db, sinr_db + n);
  return GetEesmEffectiveSinr(v);
}

extern "C" int ref_get_rbg_size(int nof_prb) {
  try {
    return get_rbg_size(nof_prb);
  } catch (const std::runtime_error&) {
    return -1;
  }
}
