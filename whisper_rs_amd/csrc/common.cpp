#include "common.h"
#include <mutex>
namespace wrs {
hipStream_t utility_stream() {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    WRS_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> g(mu);
    if (!streams[dev]) HIP_CHECK(hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking));
    return streams[dev];
}
static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
const std::string& last_error() { return g_last_error; }
void require_device(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) WRS_THROW(WRS_ENODEVICE, "no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) WRS_THROW(WRS_ENODEVICE, "device %d out of range (%d devices)", device_id, n);
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        WRS_THROW(WRS_ENODEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
}
bool g_experiment_switches = false;
const char* exp_env(const char* name) { return g_experiment_switches ? getenv(name) : nullptr; }

}  // namespace wrs
