// Wrapper translation unit: compiles the reference's stand-alone unit program
// unittest/test_effective_sinr.cpp WHERE IT LIES (it carries the reference's own copy of the AMC
// tables and of GetTBSizeFromMCS / GetMCSFromCQI / GetEesmEffectiveSinr) and exports the tables
// and functions with C linkage.  Its main() is renamed so it can live in a shared object.
#define main ref_unittest_eesm_main
#include "test_effective_sinr.cpp"
#undef main

extern "C" const int* ref_ut_tbs_table(void) { return &TransportBlockSizeTable[0][0]; }
extern "C" const int* ref_ut_mcs_to_itbs(void) { return McsToItbs; }
extern "C" const int* ref_ut_cqi_to_mcs(void) { return MapCQIToMCS; }
extern "C" const double* ref_ut_sinr_for_cqi(void) { return SINRForCQIIndex; }
extern "C" int ref_ut_tbs(int mcs, int nb_rbs) { return GetTBSizeFromMCS(mcs, nb_rbs); }
extern "C" double ref_ut_eesm(const double* s, int n) {
  std::vector<double> v(s, s + n);
  return GetEesmEffectiveSinr(v);
}
