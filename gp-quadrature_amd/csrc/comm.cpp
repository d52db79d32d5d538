// Sum all-reduce over RCCL behind the C ABI (efgp_comm_init / allreduce_sum / destroy): what the sharded
// fit needs between its spread pass and the replicated solve -- the gridded partial sums F*y and v (efgpnd.py:118-124)
// and the N-length scalars of the hyper-gradient (efgpnd.py:163, 170, 239) -- for callers that do not bring a
// torch.distributed process group.  One communicator per process (one process per GPU); the 128-byte unique id is made
// by rank 0 and handed to the other ranks by the caller's own channel (file, socket, MPI, a TCP store).
#include <rccl/rccl.h>

#include <cstring>

#include "common.hpp"

struct efgp_comm_s {
    ncclComm_t comm = nullptr;
    int device = 0;
    int rank = 0;
    int world = 1;
};

using namespace efgp;

#define EFGP_NCCL_CHECK(expr)                                                                              \
    do {                                                                                                   \
        ncclResult_t r__ = (expr);                                                                         \
        if (r__ != ncclSuccess) {                                                                          \
            set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r__), __FILE__, __LINE__);        \
            return EFGP_EHIP;                                                                              \
        }                                                                                                  \
    } while (0)

extern "C" {

int efgp_comm_unique_id(void* id_out_128_bytes) {
    EFGP_REQUIRE(id_out_128_bytes, "efgp_comm_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    EFGP_NCCL_CHECK(ncclGetUniqueId(&id));
    std::memcpy(id_out_128_bytes, &id, sizeof(id));
    return EFGP_OK;
}

int efgp_comm_init(efgp_comm_t** comm_out, int device, int rank, int world_size, const void* unique_id_128_bytes) {
    EFGP_REQUIRE(comm_out && unique_id_128_bytes, "efgp_comm_init: null argument");
    EFGP_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "efgp_comm_init: rank %d outside world of %d", rank, world_size);
    if (!device_ctx(device)) return EFGP_EHIP;
    DeviceGuard guard(device);
    ncclUniqueId id;
    std::memcpy(&id, unique_id_128_bytes, sizeof(id));
    auto* c = new efgp_comm_s();
    c->device = device;
    c->rank = rank;
    c->world = world_size;
    ncclResult_t r = ncclCommInitRank(&c->comm, world_size, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete c;
        return EFGP_EHIP;
    }
    *comm_out = c;
    return EFGP_OK;
}

int efgp_comm_allreduce_sum(efgp_comm_t* comm, double* buf, size_t n_doubles, void* stream) {
    EFGP_REQUIRE(comm && (buf || n_doubles == 0), "efgp_comm_allreduce_sum: null argument");
    if (n_doubles == 0) return EFGP_OK;
    DeviceGuard guard(comm->device);
    EFGP_NCCL_CHECK(ncclAllReduce(buf, buf, n_doubles, ncclDouble, ncclSum, comm->comm, (hipStream_t)stream));
    return EFGP_OK;
}

int efgp_comm_allreduce_minmax(efgp_comm_t* comm, double* buf, size_t n_doubles, int take_max, void* stream) {
    EFGP_REQUIRE(comm && (buf || n_doubles == 0), "efgp_comm_allreduce_minmax: null argument");
    if (n_doubles == 0) return EFGP_OK;
    DeviceGuard guard(comm->device);
    EFGP_NCCL_CHECK(ncclAllReduce(buf, buf, n_doubles, ncclDouble, take_max ? ncclMax : ncclMin, comm->comm, (hipStream_t)stream));
    return EFGP_OK;
}

int efgp_comm_broadcast(efgp_comm_t* comm, void* buf, size_t nbytes, int root, void* stream) {
    EFGP_REQUIRE(comm && (buf || nbytes == 0), "efgp_comm_broadcast: null argument");
    if (nbytes == 0) return EFGP_OK;
    DeviceGuard guard(comm->device);
    EFGP_NCCL_CHECK(ncclBroadcast(buf, buf, nbytes, ncclChar, root, comm->comm, (hipStream_t)stream));
    return EFGP_OK;
}

int efgp_comm_destroy(efgp_comm_t* comm) {
    if (!comm) return EFGP_OK;
    DeviceGuard guard(comm->device);
    if (comm->comm) (void)ncclCommDestroy(comm->comm);
    delete comm;
    return EFGP_OK;
}

}  // extern "C"
