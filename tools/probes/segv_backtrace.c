// Debug aid for GPU-box probes: on SIGSEGV / SIGABRT print the native frames of the faulting thread to stderr, then
// re-raise.  gcc -shared -fPIC -O1 -g -o segv_backtrace.so segv_backtrace.c ; loaded with ctypes, install() once.
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction old_segv, old_abrt;

static void handler(int sig) {
    void* frames[96];
    const char* head = sig == SIGSEGV ? "\n== SIGSEGV, native frames of the faulting thread ==\n" : "\n== SIGABRT, native frames ==\n";
    if (write(2, head, strlen(head)) < 0) return;
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    sigaction(sig, sig == SIGSEGV ? &old_segv : &old_abrt, 0);   // Python's faulthandler (or the default action) next
    raise(sig);
}

void install(void) {
    void* warm[2];
    backtrace(warm, 2);          // loads libgcc now, not inside the handler
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = handler;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGABRT, &sa, &old_abrt);
}
