// Sanitizer build only (tools/sanitize_cpu.sh): the HOST side of the library -- rs_api.cpp + rs_jit.cpp compiled by g++ with
// -fsanitize=address,undefined -- has no device code, so the five kernel launchers of rs_kernels.hip resolve to these refusals.
// Nothing here computes anything: without a GPU every entry point that would launch has already returned RS_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
struct RsLaunch;
extern "C" hipError_t rs_launch_cells(const RsLaunch*, int, hipStream_t) { return hipErrorNotSupported; }
extern "C" hipError_t rs_prepare_kernels(int) { return hipErrorNotSupported; }
extern "C" hipError_t rs_launch_synth(uint8_t*, int64_t, int, int, int, int, int, uint64_t, int64_t, const uint32_t*, hipStream_t) { return hipErrorNotSupported; }
extern "C" hipError_t rs_launch_copy_probe(const void*, void*, size_t, hipStream_t) { return hipErrorNotSupported; }
extern "C" hipError_t rs_launch_slice_bytes(const int64_t*, const uint8_t*, int, int, int, unsigned long long*, hipStream_t) { return hipErrorNotSupported; }
