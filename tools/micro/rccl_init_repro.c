/* rccl_init_repro.c — the smallest program that does what a Node rank does before its first step: dlopen an RCCL,
 * ncclGetUniqueId, hipSetDevice, ncclCommInitRank(nranks = 1), ncclCommDestroy.  tests/test_node_shim.py saw the one-rank
 * ncclCommInitRank of the SYSTEM RCCL (a node process has no torch) intermittently never return on this GPU pool; this
 * program takes libfluid_hip.so, node and pytest out of the picture.  Run by tools/rccl_init_probe.py with NCCL_DEBUG=INFO
 * NCCL_DEBUG_SUBSYS=INIT,ENV,NET; prints a marker before and after every call so that a hang is located.
 *   gcc -O1 -o rccl_init_repro rccl_init_repro.c -ldl      (no HIP / RCCL headers needed)
 *   ./rccl_init_repro [librccl path] [libamdhip64 path]                                                                          */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int (*fn_uid)(ncclUniqueId *);
typedef int (*fn_init)(void **, int, ncclUniqueId, int);
typedef int (*fn_destroy)(void *);
typedef int (*fn_ver)(int *);
typedef int (*fn_setdev)(int);
typedef int (*fn_count)(int *);

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
#define MARK(...) do { printf("[repro %8.3f] ", now() - t0); printf(__VA_ARGS__); printf("\n"); fflush(stdout); } while (0)

int main(int argc, char **argv)
{
    const double t0 = now();
    const char *rccl = argc > 1 ? argv[1] : "librccl.so.1";
    const char *hip = argc > 2 ? argv[2] : "libamdhip64.so";
    MARK("dlopen %s", hip);
    void *hh = dlopen(hip, RTLD_NOW | RTLD_GLOBAL);
    if (!hh) { MARK("dlopen failed: %s", dlerror()); return 2; }
    MARK("dlopen %s", rccl);
    void *h = dlopen(rccl, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { MARK("dlopen failed: %s", dlerror()); return 2; }
    Dl_info info;
    fn_uid uid = (fn_uid)dlsym(h, "ncclGetUniqueId");
    fn_init init = (fn_init)dlsym(h, "ncclCommInitRank");
    fn_destroy destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    fn_ver ver = (fn_ver)dlsym(h, "ncclGetVersion");
    fn_setdev setdev = (fn_setdev)dlsym(hh, "hipSetDevice");
    fn_count count = (fn_count)dlsym(hh, "hipGetDeviceCount");
    if (!uid || !init || !destroy || !setdev || !count) { MARK("missing symbol"); return 2; }
    if (dladdr((void *)init, &info)) MARK("ncclCommInitRank from %s", info.dli_fname);
    if (dladdr((void *)setdev, &info)) MARK("hipSetDevice from %s", info.dli_fname);
    int v = 0, n = 0, rc;
    if (ver) { ver(&v); MARK("ncclGetVersion %d", v); }
    rc = count(&n);
    MARK("hipGetDeviceCount rc %d n %d", rc, n);
    rc = setdev(0);
    MARK("hipSetDevice(0) rc %d", rc);
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    MARK("ncclGetUniqueId ...");
    rc = uid(&id);
    MARK("ncclGetUniqueId rc %d", rc);
    void *comm = 0;
    MARK("ncclCommInitRank(nranks 1, rank 0) ...");
    rc = init(&comm, 1, id, 0);
    MARK("ncclCommInitRank rc %d", rc);
    if (rc == 0) {
        rc = destroy(comm);
        MARK("ncclCommDestroy rc %d", rc);
    }
    MARK("done");
    return rc;
}
