/* ORACLE-side test infrastructure, not product code.
 *
 * LD_PRELOAD shim for the CPU-baseline worker processes of bench.py (cpu_baseline leg C, oracle/cpu_fanout_worker.py):
 * a CPU worker imports the ROCm build of torch, and the first C++-side "is a GPU there?" probe inside a forward initialises
 * the HSA runtime, which opens /dev/kfd even when no device is visible.  A GPU box admits only a handful of processes on its
 * card, so a pool of int(0.75 x cores) such workers (GEN:190) would be killed by the box's process guard.  With this shim the
 * device nodes simply do not exist for the worker: open() of /dev/kfd or /dev/dri/... fails with ENOENT, the runtime reports
 * "no device", and the worker runs on host cores only — which is all it is for.
 *
 *   gcc -shared -fPIC -O2 -o oracle/_build/libnogpu.so oracle/nogpu_shim.c -ldl        (done by __graft_entry__.build())
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <string.h>
#include <sys/types.h>

static int blocked(const char* p) {
    return p && (strncmp(p, "/dev/kfd", 8) == 0 || strncmp(p, "/dev/dri/", 9) == 0);
}

#define MODE_ARG(flags, mode)                 \
    mode_t mode = 0;                          \
    if ((flags) & (O_CREAT | O_TMPFILE)) {    \
        va_list ap;                           \
        va_start(ap, flags);                  \
        mode = (mode_t)va_arg(ap, int);       \
        va_end(ap);                           \
    }

int open(const char* path, int flags, ...) {
    static int (*real)(const char*, int, ...) = 0;
    if (blocked(path)) { errno = ENOENT; return -1; }
    if (!real) real = (int (*)(const char*, int, ...))dlsym(RTLD_NEXT, "open");
    MODE_ARG(flags, mode)
    return real(path, flags, mode);
}

int open64(const char* path, int flags, ...) {
    static int (*real)(const char*, int, ...) = 0;
    if (blocked(path)) { errno = ENOENT; return -1; }
    if (!real) real = (int (*)(const char*, int, ...))dlsym(RTLD_NEXT, "open64");
    MODE_ARG(flags, mode)
    return real(path, flags, mode);
}

int openat(int dirfd, const char* path, int flags, ...) {
    static int (*real)(int, const char*, int, ...) = 0;
    if (blocked(path)) { errno = ENOENT; return -1; }
    if (!real) real = (int (*)(int, const char*, int, ...))dlsym(RTLD_NEXT, "openat");
    MODE_ARG(flags, mode)
    return real(dirfd, path, flags, mode);
}

int openat64(int dirfd, const char* path, int flags, ...) {
    static int (*real)(int, const char*, int, ...) = 0;
    if (blocked(path)) { errno = ENOENT; return -1; }
    if (!real) real = (int (*)(int, const char*, int, ...))dlsym(RTLD_NEXT, "openat64");
    MODE_ARG(flags, mode)
    return real(dirfd, path, flags, mode);
}
