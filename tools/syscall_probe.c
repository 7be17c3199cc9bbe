/* tools/syscall_probe.c -- what the host side of the pooler pays on this box: clock_gettime, a raw syscall, a futex
 * ping-pong between two PROCESSES over a shared mapping (round trip and CPU seconds used), a shared robust mutex. */
#define _GNU_SOURCE
#include <linux/futex.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static double cpu(void) { struct rusage a, b; getrusage(RUSAGE_SELF, &a); getrusage(RUSAGE_CHILDREN, &b);
    return a.ru_utime.tv_sec + a.ru_utime.tv_usec * 1e-6 + a.ru_stime.tv_sec + a.ru_stime.tv_usec * 1e-6 + b.ru_utime.tv_sec + b.ru_utime.tv_usec * 1e-6 + b.ru_stime.tv_sec + b.ru_stime.tv_usec * 1e-6; }
int main(void) {
    const int N = 1000000;
    double t0 = now(); volatile double s = 0;
    for (int i = 0; i < N; i++) s += now();
    printf("clock_gettime: %.1f ns per call\n", (now() - t0) / N * 1e9);
    t0 = now();
    for (int i = 0; i < N; i++) syscall(SYS_getppid);
    printf("raw syscall (getppid): %.1f ns per call\n", (now() - t0) / N * 1e9);
    uint32_t *w = mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    w[0] = w[1] = 0;
    const int R = 100000;
    double c0 = cpu(); t0 = now();
    pid_t p = fork();
    if (p == 0) {
        for (int i = 0; i < R; i++) {
            while (__atomic_load_n(&w[0], __ATOMIC_ACQUIRE) != (uint32_t)(i + 1)) syscall(SYS_futex, &w[0], FUTEX_WAIT, i, NULL, NULL, 0);
            __atomic_store_n(&w[1], i + 1, __ATOMIC_RELEASE); syscall(SYS_futex, &w[1], FUTEX_WAKE, 1, NULL, NULL, 0);
        }
        _exit(0);
    }
    for (int i = 0; i < R; i++) {
        __atomic_store_n(&w[0], i + 1, __ATOMIC_RELEASE); syscall(SYS_futex, &w[0], FUTEX_WAKE, 1, NULL, NULL, 0);
        while (__atomic_load_n(&w[1], __ATOMIC_ACQUIRE) != (uint32_t)(i + 1)) syscall(SYS_futex, &w[1], FUTEX_WAIT, i, NULL, NULL, 0);
    }
    waitpid(p, NULL, 0);
    printf("futex ping-pong between two processes: %.2f us per round trip, %.2f us of CPU per round trip (both sides)\n",
           (now() - t0) / R * 1e6, (cpu() - c0) / R * 1e6);
    /* 32 processes taking a process-shared robust adaptive mutex 20000 times each, 1 us of work inside */
    pthread_mutex_t *m = (pthread_mutex_t *)(w + 64); pthread_mutexattr_t ma; pthread_mutexattr_init(&ma);
    pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED); pthread_mutexattr_setrobust(&ma, PTHREAD_MUTEX_ROBUST);
    pthread_mutexattr_settype(&ma, PTHREAD_MUTEX_ADAPTIVE_NP); pthread_mutex_init(m, &ma);
    volatile uint64_t *ctr = (uint64_t *)(w + 256);
    for (int np = 4; np <= 64; np *= 4) {
        c0 = cpu(); t0 = now(); *ctr = 0;
        for (int i = 0; i < np; i++) if (fork() == 0) { for (int j = 0; j < 20000; j++) { pthread_mutex_lock(m); (*ctr)++; pthread_mutex_unlock(m); } _exit(0); }
        for (int i = 0; i < np; i++) wait(NULL);
        printf("shared robust mutex, %2d processes x 20000: %.2f us per acquisition (wall), %.2f us of CPU per acquisition\n", np,
               (now() - t0) / (np * 20000.0) * 1e6, (cpu() - c0) / (np * 20000.0) * 1e6);
    }
    return 0;
}
