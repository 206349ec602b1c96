// Wrapper translation unit: compiles the reference's stand-alone unit program
// unittest/test_tp_algos.cpp WHERE IT LIES and exports its static MaximizeCell / VogelApproximate
// (the int-keyed twins of downlink-transport-scheduler.cpp:351-376 / :378-451, same std::sort call,
// same comparator, same greedy scan) with C linkage.  main() is renamed.
#define main ref_tp_algos_main
#include "test_tp_algos.cpp"
#undef main

static int** rows_of(const int* grid, int R, int S) {
  int** p = new int*[R];
  for (int i = 0; i < R; i++) p[i] = const_cast<int*>(grid + (long)i * S);
  return p;
}

extern "C" void ref_maximize_cell_int(const int* grid, const int* quota, int R, int S, int* rbg_to_slice) {
  int** rows = rows_of(grid, R, S);
  vector<int> q(quota, quota + S);
  vector<int> out = MaximizeCell(rows, q, R, S);
  for (int i = 0; i < R; i++) rbg_to_slice[i] = out[i];
  delete[] rows;
}

extern "C" void ref_vogel_int(const int* grid, const int* quota, int R, int S, int* rbg_to_slice) {
  int** rows = rows_of(grid, R, S);
  vector<int> q(quota, quota + S);
  vector<int> out = VogelApproximate(rows, q, R, S);
  for (int i = 0; i < R; i++) rbg_to_slice[i] = out[i];
  delete[] rows;
}
