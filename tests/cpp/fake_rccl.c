/* fake_rccl.c -- TEST INFRASTRUCTURE: an RCCL-shaped library for tests/test_rccl_group.py.  It exports the symbols
 * csrc/rt_rccl.h binds (ncclCommInitAll, ncclCommDestroy, ncclCommAbort, ncclGroupStart, ncclGroupEnd, ncclSend,
 * ncclRecv, ncclGetErrorString), moves no data, and fails on request:
 *   FAKE_RCCL_FAIL_SEND=k   the k-th ncclSend (1-based) of the process returns ncclInternalError
 *   FAKE_RCCL_FAIL_RECV=k   the k-th ncclRecv
 *   FAKE_RCCL_FAIL_START=1  ncclGroupStart fails (no group is opened)
 *   FAKE_RCCL_FAIL_END=1    ncclGroupEnd fails (the group is closed all the same)
 * fake_rccl_state(out[8]) = {group depth, sends, recvs, live communicators, aborts, destroys, group starts, group ends}. */
#include <stdlib.h>
#include <string.h>

typedef int ncclResult_t;
typedef struct fake_comm { int rank; int live; } *ncclComm_t;
enum { ncclSuccess = 0, ncclInternalError = 3 };

static int g_depth, g_sends, g_recvs, g_live, g_aborts, g_destroys, g_starts, g_ends;

static int knob(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
    (void)devs;
    for (int i = 0; i < n; ++i) {
        comms[i] = (ncclComm_t)malloc(sizeof(struct fake_comm));
        comms[i]->rank = i;
        comms[i]->live = 1;
        ++g_live;
    }
    return ncclSuccess;
}
static ncclResult_t end_comm(ncclComm_t c, int* counter) {
    if (!c || !c->live) return ncclInternalError;
    c->live = 0;
    --g_live;
    ++*counter;
    free(c);
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { return end_comm(c, &g_destroys); }
ncclResult_t ncclCommAbort(ncclComm_t c) { return end_comm(c, &g_aborts); }
ncclResult_t ncclGroupStart(void) {
    if (knob("FAKE_RCCL_FAIL_START")) return ncclInternalError;
    ++g_depth;
    ++g_starts;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd(void) {
    if (g_depth <= 0) return ncclInternalError;
    --g_depth;
    ++g_ends;
    return knob("FAKE_RCCL_FAIL_END") ? ncclInternalError : ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, int type, int peer, ncclComm_t comm, void* stream) {
    (void)buf; (void)count; (void)type; (void)peer; (void)stream;
    if (g_depth <= 0 || !comm || !comm->live) return ncclInternalError;
    ++g_sends;
    return g_sends == knob("FAKE_RCCL_FAIL_SEND") ? ncclInternalError : ncclSuccess;
}
ncclResult_t ncclRecv(void* buf, size_t count, int type, int peer, ncclComm_t comm, void* stream) {
    (void)buf; (void)count; (void)type; (void)peer; (void)stream;
    if (g_depth <= 0 || !comm || !comm->live) return ncclInternalError;
    ++g_recvs;
    return g_recvs == knob("FAKE_RCCL_FAIL_RECV") ? ncclInternalError : ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake internal error"; }

void fake_rccl_state(int out[8]) {
    out[0] = g_depth; out[1] = g_sends; out[2] = g_recvs; out[3] = g_live;
    out[4] = g_aborts; out[5] = g_destroys; out[6] = g_starts; out[7] = g_ends;
}
