// Hardware-queue probe: a one-wavefront kernel that runs for a requested wall time.
//
// HIP multiplexes its streams onto a handful of hardware (AQL) queues; two streams that land on one queue execute
// their kernels back to back however independent they are.  The stream-overlapped schedule of the G+D iteration
// (pdgn_amd/trainer.py) needs the default stream's queue to itself, so pdgn_amd/streams.py times pairs of these
// kernels on candidate streams: concurrent -> different queues, serialised -> same queue.
#include <hip/hip_runtime.h>
#include "../../include/pdgn_hip.h"

__global__ void spin_kernel(unsigned long long ticks, unsigned long long *sink) {
    unsigned long long t0 = wall_clock64();                    // constant-rate (100 MHz) counter
    unsigned long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (sink && t == 0) *sink = t;                             // keeps the loop observable
}

extern "C" int pdgn_spin(unsigned int microseconds, pdgn_stream_t stream) {
    if (microseconds > 100000u) return PDGN_ERR_INVALID;          // a probe, not a sleep: 100 ms at most
    int rate_khz = 100000;                                     // wall_clock64 rate; queried so a part that differs is handled
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, dev) == hipSuccess && v > 0) rate_khz = v;
    }
    unsigned long long ticks = (unsigned long long)microseconds * (unsigned long long)rate_khz / 1000ull;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks, (unsigned long long *)nullptr);
    return (int)hipGetLastError();
}
