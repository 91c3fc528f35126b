// papr_readbatch.h — the unit the ingest waits on: the reads of one staging chunk (reader threads: papr_runtime_internal.h;
// io_uring: papr_uring.h).
#ifndef PAPR_READBATCH_H
#define PAPR_READBATCH_H
namespace papr_rt {
struct ReadBatch {
    int pending = 0;
    int error = 0;
};
}  // namespace papr_rt
#endif
