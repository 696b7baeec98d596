// Can the host store directly into device memory (large BAR), and how long does a kernel that
// polls such a word take to see it?  (scripts/dev: a probe, not product code)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csignal>
#include <thread>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_echo(volatile unsigned *mailbox, volatile unsigned *reply, unsigned rounds, unsigned long long *spins_out) {
    unsigned long long spins = 0;
    for (unsigned r = 1; r <= rounds; ++r) {
        unsigned long long guard = 0;
        while (__hip_atomic_load((unsigned *) mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < r) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if (++guard > 400000000ull) { *reply = 0xDEADu; return; }
        }
        __hip_atomic_store((unsigned *) reply, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *spins_out = spins;
}

static void on_segv(int) { printf("SIGSEGV on host access to device memory: no large-BAR access\n"); fflush(stdout); _exit(3); }

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0: fine-grained device memory, 1: plain hipMalloc, 2: pinned host (baseline)
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    int large_bar = -1;
    (void) hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("mode %d, hipDeviceAttributeIsLargeBar = %d\n", mode, large_bar);
    unsigned *mailbox = nullptr, *reply = nullptr;
    unsigned long long *spins = nullptr;
    if (mode == 0) CK(hipExtMallocWithFlags((void **) &mailbox, 4096, hipDeviceMallocFinegrained));
    else if (mode == 1) CK(hipMalloc((void **) &mailbox, 4096));
    else CK(hipHostMalloc((void **) &mailbox, 4096, hipHostMallocDefault));
    CK(hipHostMalloc((void **) &reply, 4096, hipHostMallocDefault));
    CK(hipHostMalloc((void **) &spins, 4096, hipHostMallocDefault));
    CK(hipMemset(mailbox, 0, 4096));
    memset(reply, 0, 4096);
    CK(hipDeviceSynchronize());
    // host store
    volatile unsigned *mb = mailbox;
    *mb = 0;  // faults here without large-BAR access
    printf("host store to the mailbox worked\n");
    const unsigned rounds = 2000;
    hipLaunchKernelGGL(k_echo, dim3(1), dim3(64), 0, 0, mailbox, reply, rounds, spins);
    CK(hipGetLastError());
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    double worst = 0, sum = 0;
    for (unsigned r = 1; r <= rounds; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        *mb = r;
        __sync_synchronize();
        while (*(volatile unsigned *) reply != r) {
            if (*(volatile unsigned *) reply == 0xDEADu) { printf("kernel gave up\n"); return 2; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { printf("timeout at round %u\n", r); *mb = rounds + 1; return 2; }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        sum += us;
        if (us > worst) worst = us;
    }
    CK(hipDeviceSynchronize());
    printf("round trip host store -> kernel sees it -> kernel's reply seen by the host: mean %.2f us, worst %.2f us (%u rounds, %llu sleeps)\n",
           sum / rounds, worst, rounds, *spins);
    return 0;
}
