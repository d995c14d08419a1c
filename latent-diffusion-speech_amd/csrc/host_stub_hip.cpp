// Host stand-in for the HIP runtime calls liblds makes at *_create / planning time, for the sanitizer build only (make asan):
// "device" memory is host memory, so the weight packers, the split-bf16 twin packer (which reads weights back), the workspace
// planners and the argument validation run under AddressSanitizer / UBSan on a machine without a GPU.  Nothing here can launch a
// kernel: every launch entry point fails, as does everything about events and streams that is not a no-op.  TEST INFRASTRUCTURE --
// never linked into liblds.so.
#include <hip/hip_runtime_api.h>

#include <stdlib.h>
#include <string.h>

extern "C" {
hipError_t hipMalloc(void** p, size_t n) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s_, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memcpy((char*)d + r * dpitch, (const char*)s_ + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t) {
    for (size_t i = 0; i < count; ++i) memcpy((char*)d + 4 * i, &v, 4);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "host stub"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipExtLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t, hipEvent_t, hipEvent_t, int) { return hipErrorNotSupported; }
hipError_t hipEventCreate(hipEvent_t*) { return hipErrorNotSupported; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipErrorNotSupported; }
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t) { return hipErrorNotSupported; }
void** __hipRegisterFatBinary(const void*) {
    static void* handle = nullptr;
    return &handle;
}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned int, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3*, dim3*, size_t*, hipStream_t*) { return hipErrorNotSupported; }
}
