#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gptr;
// every lane fetches element idx[lane] (arbitrary) into LDS slot [wave*64 + lane]
__global__ void k(const uint4* in, const unsigned* idx, uint4* out, int rounds) {
    __shared__ uint4 stage[2][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r = 0; r < rounds; r++) {
        unsigned i = idx[(blockIdx.x * rounds + r) * 256 + threadIdx.x];
        if (threadIdx.x % 3 != 1) { // some lanes inactive
            __builtin_amdgcn_global_load_lds((gptr)(in + i), (lds_ptr)(&stage[0][wave * 64]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(in + i + 1), (lds_ptr)(&stage[1][wave * 64]), 16, 0, 0);
        }
        uint4 a = stage[0][threadIdx.x], b = stage[1][threadIdx.x];
        if (threadIdx.x % 3 == 1) a = b = make_uint4(7, 7, 7, 7);
        out[((blockIdx.x * rounds + r) * 256 + threadIdx.x) * 2] = a;
        out[((blockIdx.x * rounds + r) * 256 + threadIdx.x) * 2 + 1] = b;
    }
}
int main() {
    const int blocks = 64, rounds = 8, n = blocks * rounds * 256, m = 1 << 16;
    std::vector<uint4> in(m + 1);
    for (int i = 0; i <= m; i++) in[i] = make_uint4(i, i * 3, i ^ 0x5555, ~i);
    std::vector<unsigned> idx(n);
    unsigned s = 12345;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; idx[i] = (s >> 8) % m; }
    uint4 *din, *dout; unsigned* didx;
    hipMalloc(&din, in.size() * 16); hipMalloc(&dout, n * 32); hipMalloc(&didx, n * 4);
    hipMemcpy(din, in.data(), in.size() * 16, hipMemcpyHostToDevice);
    hipMemcpy(didx, idx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, din, didx, dout, rounds);
    std::vector<uint4> out(2 * n);
    hipMemcpy(out.data(), dout, n * 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        bool skip = (i % 256) % 3 == 1;
        uint4 wa = skip ? make_uint4(7, 7, 7, 7) : in[idx[i]], wb = skip ? make_uint4(7, 7, 7, 7) : in[idx[i] + 1];
        uint4 a = out[2 * i], b = out[2 * i + 1];
        if (a.x != wa.x || a.y != wa.y || a.z != wa.z || a.w != wa.w || b.x != wb.x || b.w != wb.w) bad++;
    }
    printf("mismatches %d of %d (%s)\n", bad, n, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
