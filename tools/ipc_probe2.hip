// tools/ipc_probe2.hip -- does an importer poison the exporter's address?  Exporter (parent) and importer (child, forked
// BEFORE any HIP call) talk over pipes.  Scenarios: the importer closes its mapping before / after / never relative to
// the exporter's hipFree; then the exporter allocates again (usually the same address) and exports again.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <signal.h>
struct Msg { int op; hipIpcMemHandle_t h; };   // op 1: open + reply, 2: close + reply, 3: exit, 4: exit without closing
static void importer(int rd, int wr) {
    void *p = nullptr; Msg m; int ok;
    while (read(rd, &m, sizeof m) == (ssize_t)sizeof m) {
        if (m.op == 1) { hipError_t e = hipIpcOpenMemHandle(&p, m.h, hipIpcMemLazyEnablePeerAccess); ok = e == hipSuccess; if (!ok) { (void)hipGetLastError(); p = nullptr; } write(wr, &ok, sizeof ok); }
        else if (m.op == 2) { ok = p ? hipIpcCloseMemHandle(p) == hipSuccess : 0; p = nullptr; write(wr, &ok, sizeof ok); }
        else if (m.op == 3) { if (p) hipIpcCloseMemHandle(p); _exit(0); }
        else _exit(0);
    }
    _exit(0);
}
static int to_imp, from_imp; static pid_t imp;
static void start_importer() { int a[2], b[2]; pipe(a); pipe(b); imp = fork(); if (imp == 0) { close(a[1]); close(b[0]); importer(a[0], b[1]); } close(a[0]); close(b[1]); to_imp = a[1]; from_imp = b[0]; }
static int ask(int op, hipIpcMemHandle_t *h) { Msg m; memset(&m, 0, sizeof m); m.op = op; if (h) m.h = *h; write(to_imp, &m, sizeof m); if (op >= 3) { waitpid(imp, nullptr, 0); return 1; } int ok = 0; read(from_imp, &ok, sizeof ok); return ok; }
int main(int argc, char **argv) {
    size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : 20000;
    const char *names[] = {"importer closes, then exporter frees", "exporter frees while the importer has it open, importer closes later",
                           "importer exits WITHOUT closing, then exporter frees", "exporter frees, importer is killed (SIGKILL) with it open"};
    for (int sc = 0; sc < 4; sc++) {
        start_importer();   // forked before this process touches HIP the first time in scenario 0; later forks happen with HIP live in the parent (the child only uses what it opens itself)
        void *p = nullptr; hipIpcMemHandle_t h;
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
        hipError_t e = hipIpcGetMemHandle(&h, p);
        printf("scenario %d (%s): first export of %p: %s\n", sc, names[sc], p, hipGetErrorString(e));
        int opened = ask(1, &h);
        printf("   importer opened it: %d\n", opened);
        if (sc == 0) { ask(2, nullptr); hipFree(p); ask(3, nullptr); }
        if (sc == 1) { hipFree(p); }
        if (sc == 2) { ask(4, nullptr); hipFree(p); }
        if (sc == 3) { hipFree(p); kill(imp, SIGKILL); waitpid(imp, nullptr, 0); }
        int fails = 0; void *q = nullptr;
        for (int i = 0; i < 10; i++) {
            if (hipMalloc(&q, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
            hipIpcMemHandle_t h2; hipError_t e2 = hipIpcGetMemHandle(&h2, q);
            if (e2 != hipSuccess) { (void)hipGetLastError(); fails++; }
            if (i == 0) printf("   again: %p%s -> %s\n", q, q == p ? " (the SAME address)" : "", hipGetErrorString(e2));
            hipFree(q);
        }
        printf("   %d of 10 re-exports failed\n", fails);
        if (sc == 1) { ask(2, nullptr); ask(3, nullptr); }
    }
    return 0;
}
