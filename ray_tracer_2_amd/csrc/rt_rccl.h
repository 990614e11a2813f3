// rt_rccl.h -- the RCCL side of rt_render_multi (rt_api.hip): the library is loaded with dlopen on first
// multi-device use, and the per-frame gather is ONE group of ncclSend / ncclRecv pairs over xGMI.
//
// Error rule of the group: whatever fails between ncclGroupStart and ncclGroupEnd, the group is CLOSED before the
// function returns (a return in between would leave every later RCCL call of the process inside an open group), and
// the caller aborts the communicators of a failed gather (a half-issued group can leave a send without its receive
// on a stream).  Kept free of HIP calls -- device switching goes through a callback -- so that
// tests/test_rccl_group.py can drive it on the CPU against a stub library that fails the k-th send
// (tests/cpp/fake_rccl.c).
#ifndef RT_RCCL_H
#define RT_RCCL_H

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only

#include <string>
#include <vector>

namespace rtd {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: ncclCommDestroy stands in
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;

    // `path`: one library to load (tests); null = the installed librccl
    bool load(const char* path = nullptr) {
        if (lib) return true;
        if (path) {
            lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
            }
        }
        if (!lib) {
            const char* e = dlerror();
            why = std::string("cannot load librccl: ") + (e ? e : "?");
            return false;
        }
        auto sym = [&](const char* n, bool required = true) {
            void* f = dlsym(lib, n);
            if (!f && required) why = std::string("librccl lacks ") + n;
            return f;
        };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort", false);
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
    void unload() {
        if (lib) dlclose(lib);
        *this = RcclApi();
    }
    // end a communicator whose last group failed (or any communicator): abort where the library has it
    void drop(ncclComm_t c) const { (void)(CommAbort ? CommAbort(c) : CommDestroy(c)); }
};

// One rank's share of the gather: `count` floats from the rank's image (on its device and stream, through its
// communicator) into the root's buffer (root's device, stream and communicator: rank 0's).
struct GatherLeg {
    const void* send_buf;
    void* recv_buf;
    size_t count;  // floats; 0 = nothing to send
    int rank;
    int device;
    ncclComm_t comm;
    hipStream_t stream;
};

// The gather as one group.  set_device(device) makes a device current (each call is issued with its communicator's
// device current: older RCCL releases require it in one-thread use); it returns 0 or an error it has described in
// `err`.  Returns true when every call of the group -- and the group itself -- succeeded; otherwise `err` names the
// first failure.  The group is closed on EVERY path.
template <class SetDevice>
bool rccl_grouped_gather(const RcclApi& api, const std::vector<GatherLeg>& legs, int root_device, ncclComm_t root_comm,
                         hipStream_t root_stream, SetDevice&& set_device, std::string& err) {
    err.clear();
    ncclResult_t r = api.GroupStart();
    if (r != ncclSuccess) {  // (no group is open)
        err = std::string("ncclGroupStart: ") + api.GetErrorString(r);
        return false;
    }
    auto note = [&](const char* what, ncclResult_t res) {
        if (res != ncclSuccess && err.empty()) err = std::string(what) + ": " + api.GetErrorString(res);
        return res == ncclSuccess;
    };
    for (const GatherLeg& g : legs) {
        if (g.count == 0) continue;
        if (set_device(g.device, err) != 0) break;
        if (!note("ncclSend", api.Send(g.send_buf, g.count, ncclFloat, 0, g.comm, g.stream))) break;
        if (set_device(root_device, err) != 0) break;
        if (!note("ncclRecv", api.Recv(g.recv_buf, g.count, ncclFloat, g.rank, root_comm, root_stream))) break;
    }
    r = api.GroupEnd();  // (always: see the error rule above)
    if (err.empty()) {
        note("ncclGroupEnd", r);
    }
    std::string ignored;
    (void)set_device(root_device, ignored);
    return err.empty();
}

}  // namespace rtd

#endif
