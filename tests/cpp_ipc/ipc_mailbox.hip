// ipc_mailbox.hip -- what wm_shard.hip's mailboxes rest on, between two PROCESSES (one GPU is enough): uncached device
// memory from hipExtMallocWithFlags exported with hipIpcGetMemHandle and mapped with hipIpcOpenMemHandle, a kernel of
// the importing process storing tagged words into it at system scope, and a kernel of the owning process -- already
// RUNNING and polling -- seeing them.   usage: ipc_mailbox owner <file>   |   ipc_mailbox peer <file>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

constexpr int kWords = 68;

__global__ void k_send(unsigned long long *box, unsigned round) {
    if (threadIdx.x < kWords)
        __hip_atomic_store(box + threadIdx.x, ((unsigned long long) round << 32) | (1000u + threadIdx.x), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_poll(const unsigned long long *box, unsigned round, unsigned long long limit_ticks, int *ok) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if (threadIdx.x < kWords) {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            const unsigned long long v = __hip_atomic_load(box + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned) (v >> 32) == round) {
                if ((unsigned) v != 1000u + threadIdx.x) atomicAdd(&bad, 1);
                break;
            }
            if (wall_clock64() - t0 > limit_ticks) {
                atomicAdd(&bad, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *ok = bad == 0 ? 1 : 0;
}

static bool wait_for_file(const char *path, void *buf, size_t bytes, int seconds) {
    for (int k = 0; k < seconds * 100; ++k) {
        if (FILE *f = std::fopen(path, "rb")) {
            const size_t got = std::fread(buf, 1, bytes, f);
            std::fclose(f);
            if (got == bytes) return true;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    return false;
}

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    const bool owner = std::strcmp(argv[1], "owner") == 0;
    char handle_path[512], done_path[512];
    std::snprintf(handle_path, sizeof(handle_path), "%s.handle", argv[2]);
    std::snprintf(done_path, sizeof(done_path), "%s.sent", argv[2]);
    CK(hipSetDevice(0));
    if (owner) {
        void *box = nullptr;
        CK(hipExtMallocWithFlags(&box, 4096, hipDeviceMallocUncached));
        CK(hipMemset(box, 0, 4096));
        CK(hipDeviceSynchronize());
        hipIpcMemHandle_t h;
        CK(hipIpcGetMemHandle(&h, box));
        int *ok = nullptr;
        CK(hipHostMalloc((void **) &ok, sizeof(int), hipHostMallocDefault));
        *ok = -1;
        // the poller is launched FIRST and runs while the peer process maps the memory and stores into it
        hipLaunchKernelGGL(k_poll, dim3(1), dim3(128), 0, nullptr, (const unsigned long long *) box, 7u, 20ull * 100000000ull, ok);
        CK(hipGetLastError());
        char tmp[520];
        std::snprintf(tmp, sizeof(tmp), "%s.tmp", handle_path);
        FILE *f = std::fopen(tmp, "wb");
        if (!f) return 3;
        std::fwrite(&h, 1, sizeof(h), f);
        std::fclose(f);
        std::rename(tmp, handle_path);
        CK(hipDeviceSynchronize());
        std::printf("owner: poll %s\n", *ok == 1 ? "saw every word of the round" : "FAILED");
        return *ok == 1 ? 0 : 4;
    }
    hipIpcMemHandle_t h;
    if (!wait_for_file(handle_path, &h, sizeof(h), 30)) {
        std::fprintf(stderr, "peer: no handle\n");
        return 3;
    }
    void *box = nullptr;
    CK(hipIpcOpenMemHandle(&box, h, hipIpcMemLazyEnablePeerAccess));
    hipLaunchKernelGGL(k_send, dim3(1), dim3(128), 0, nullptr, (unsigned long long *) box, 7u);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipIpcCloseMemHandle(box));
    std::printf("peer: sent\n");
    return 0;
}
