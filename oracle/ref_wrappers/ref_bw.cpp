// Wrapper translation unit: compiles the reference's BandwidthManager WHERE IT LIES
// (/root/reference/src/core/spectrum/bandwidth-manager.cpp, added with -I by oracle/Makefile) to pin the PRB
// grid the schedulers see through GetDlSubChannels().size(): 100 MHz -> 512 PRBs, 20 MHz -> 100, ...
// (bandwidth-manager.cpp:38, 98-102).  No reference source is copied here.
#include "core/spectrum/bandwidth-manager.cpp"

extern "C" int ref_dl_subchannels(double bw_mhz) {
  BandwidthManager m(bw_mhz, bw_mhz, 0, 0);
  return (int)m.GetDlSubChannels().size();
}

extern "C" int ref_ul_subchannels(double bw_mhz) {
  BandwidthManager m(bw_mhz, bw_mhz, 0, 0);
  return (int)m.GetUlSubChannels().size();
}
