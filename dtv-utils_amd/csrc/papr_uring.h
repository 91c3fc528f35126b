// papr_uring.h — the ingest's second reader: one io_uring instead of a pool of threads blocked in pread.
//
// Used for files that are NOT in the page cache (read with O_DIRECT: the disk DMAs straight into the pinned staging
// buffer, there is no copy for a CPU to do) — there the reader threads only exist to keep enough requests in flight,
// which a ring does from the one thread that also issues the H2D copies and the kernels.  Files that ARE in the page
// cache stay with the thread pool: a buffered io_uring read of cached pages is a memcpy executed inline by the
// submitter, i.e. one core instead of sixteen.
//
// Raw system calls (no liburing in the image): io_uring_setup + the ring mmaps + io_uring_enter.  Single-threaded by
// construction (the ingest loop of one context).  create() returns nullptr where the kernel or a seccomp profile does
// not offer io_uring — the caller keeps the thread pool then.
#ifndef PAPR_URING_H
#define PAPR_URING_H

#include <errno.h>
#include <linux/io_uring.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <deque>
#include <vector>

#include "papr_readbatch.h"  // ReadBatch {pending, error}

namespace papr_rt {

class UringReader {
  public:
    static UringReader *create(unsigned entries)
    {
        UringReader *u = new UringReader();
        if (!u->init(entries)) {
            delete u;
            return nullptr;
        }
        return u;
    }
    ~UringReader()
    {
        abandon();
        if (sqes_ && sqes_ != MAP_FAILED)
            munmap(sqes_, sqe_bytes_);
        if (cq_ptr_ && cq_ptr_ != MAP_FAILED && cq_ptr_ != sq_ptr_)
            munmap(cq_ptr_, cq_bytes_);
        if (sq_ptr_ && sq_ptr_ != MAP_FAILED)
            munmap(sq_ptr_, sq_bytes_);
        if (fd_ >= 0)
            close(fd_);
    }

    // Queue reads of file bytes [off, off + len) into dst, in pieces of at most `piece` bytes, on behalf of `batch`.
    // fd may be an O_DIRECT descriptor (off, len, dst 4 KiB-aligned then); `fd_buffered` finishes whatever a piece
    // returns short (the unaligned end of the file) or the direct descriptor refuses.
    void submit(ReadBatch *batch, int fd, int fd_buffered, uint64_t off, uint64_t len, unsigned char *dst, uint64_t piece)
    {
        while (len) {
            const uint64_t n = len < piece ? len : piece;
            Req *r = new Req{batch, fd, fd_buffered, off, n, dst};
            batch->pending++;
            if (dead_) {  // (not used after a failure; if it is, the request fails instead of hanging)
                batch->error = batch->error ? batch->error : EIO;
                batch->pending--;
                delete r;
            } else if (!backlog_.empty() || !enqueue(r)) {
                backlog_.push_back(r);
            }
            off += n;
            len -= n;
            dst += n;
        }
        enter(0);
    }

    // Block until every request of the batch has completed; returns the batch's error (0 = none).
    int wait(ReadBatch *batch)
    {
        while (batch->pending > 0) {
            if (reap())
                continue;
            if (inflight_ == 0 && backlog_.empty()) {  // cannot happen: nothing left that could complete it
                batch->error = batch->error ? batch->error : EIO;
                batch->pending = 0;
                break;
            }
            if (!enter(1)) {
                // the ring itself failed.  Requests may still be in flight — the kernel may still be writing into the
                // staging buffers, and their completions would touch this batch: take what completes, cut the rest off
                // from every batch, and never use this ring again (the caller goes back to the reader threads)
                batch->error = batch->error ? batch->error : EIO;
                abandon();
                batch->pending = 0;
                break;
            }
        }
        return batch->error;
    }

    unsigned entries() const { return entries_; }
    bool dead() const { return dead_; }
    bool poisoned() const { return poisoned_; }  // given up with requests the kernel would not give back: its buffers are not safe

  private:
    struct Req {
        ReadBatch *batch;
        int fd, fd_buffered;
        uint64_t off, len;
        unsigned char *dst;
        size_t live_at = 0;  // position in live_ while the kernel has it
    };

    // Give the ring up: requests that were never handed to the kernel are dropped; for those the kernel has, completions
    // are taken for up to five seconds (the completion ring is shared memory: no system call needed); whatever is still
    // out then is detached from its batch (a late completion lands in `orphans_`) and leaked rather than freed under it.
    void abandon()
    {
        if (dead_ && live_.empty() && backlog_.empty())
            return;
        dead_ = true;
        for (Req *r : backlog_) {
            if (r->batch->pending > 0)
                r->batch->pending--;
            delete r;
        }
        backlog_.clear();
        // entries that sit in the submission ring but were never handed to the kernel (exactly the state a failed
        // io_uring_enter leaves behind) can never complete: they are the LAST `unsubmitted_` of live_ — taken back at once
        while (unsubmitted_ && !live_.empty()) {
            Req *r = live_.back();
            live_.pop_back();
            unsubmitted_--;
            inflight_--;
            sq_tail_->store(sq_tail_->load(std::memory_order_relaxed) - 1, std::memory_order_release);
            if (r->batch->pending > 0)
                r->batch->pending--;
            delete r;
        }
        unsubmitted_ = 0;
        for (int spin = 0; spin < 50000 && !live_.empty(); spin++) {  // only what the kernel really has is waited for
            if (!reap())
                usleep(100);
        }
        // What the kernel still has after that must not complete into staging buffers the reader threads are about to reuse.
        // close() alone does not see to that — the three ring mappings hold references to the ring's file, so the ring lives
        // on behind the closed descriptor — hence: cancel everything synchronously (IORING_REGISTER_SYNC_CANCEL, Linux 6.0+),
        // take the completions that produces, and only then let the ring go.  If requests are out even then (an older kernel,
        // a device that does not answer), the ring says so (`poisoned()`): the ingest then fails instead of reading on into
        // buffers a stray completion may still write.
        if (!live_.empty() && fd_ >= 0) {
            struct {
                uint64_t addr;
                int32_t fd;
                uint32_t flags;
                struct { int64_t tv_sec; long long tv_nsec; } timeout;
                uint8_t opcode, pad[7];
                uint64_t pad2[3];
            } reg;
            memset(&reg, 0, sizeof(reg));
            reg.flags = 1u << 2;  // IORING_ASYNC_CANCEL_ANY
            reg.timeout.tv_sec = 2;
            (void)syscall(__NR_io_uring_register, fd_, 24 /* IORING_REGISTER_SYNC_CANCEL */, &reg, 1);
            for (int spin = 0; spin < 20000 && !live_.empty(); spin++) {
                if (!reap())
                    usleep(100);
            }
        }
        poisoned_ = !live_.empty();
        for (Req *r : live_)
            r->batch = &orphans_;
        live_.clear();
        if (fd_ >= 0) {
            close(fd_);
            fd_ = -1;
        }
    }

    bool init(unsigned entries)
    {
        struct io_uring_params p;
        memset(&p, 0, sizeof(p));
        fd_ = (int)syscall(__NR_io_uring_setup, entries, &p);
        if (fd_ < 0)
            return false;
        entries_ = p.sq_entries;
        sq_bytes_ = p.sq_off.array + p.sq_entries * sizeof(uint32_t);
        cq_bytes_ = p.cq_off.cqes + p.cq_entries * sizeof(struct io_uring_cqe);
        const bool single = (p.features & IORING_FEAT_SINGLE_MMAP) != 0;
        if (single)
            sq_bytes_ = cq_bytes_ = sq_bytes_ > cq_bytes_ ? sq_bytes_ : cq_bytes_;
        sq_ptr_ = mmap(nullptr, sq_bytes_, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd_, IORING_OFF_SQ_RING);
        if (sq_ptr_ == MAP_FAILED)
            return false;
        cq_ptr_ = single ? sq_ptr_
                         : mmap(nullptr, cq_bytes_, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd_, IORING_OFF_CQ_RING);
        if (cq_ptr_ == MAP_FAILED)
            return false;
        sqe_bytes_ = p.sq_entries * sizeof(struct io_uring_sqe);
        sqes_ = (struct io_uring_sqe *)mmap(nullptr, sqe_bytes_, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd_,
                                            IORING_OFF_SQES);
        if (sqes_ == MAP_FAILED)
            return false;
        char *sq = (char *)sq_ptr_, *cq = (char *)cq_ptr_;
        sq_head_ = (std::atomic<uint32_t> *)(sq + p.sq_off.head);
        sq_tail_ = (std::atomic<uint32_t> *)(sq + p.sq_off.tail);
        sq_mask_ = *(uint32_t *)(sq + p.sq_off.ring_mask);
        sq_array_ = (uint32_t *)(sq + p.sq_off.array);
        cq_head_ = (std::atomic<uint32_t> *)(cq + p.cq_off.head);
        cq_tail_ = (std::atomic<uint32_t> *)(cq + p.cq_off.tail);
        cq_mask_ = *(uint32_t *)(cq + p.cq_off.ring_mask);
        cqes_ = (struct io_uring_cqe *)(cq + p.cq_off.cqes);
        return true;
    }

    // one request into the submission ring; false while the ring (or the completion side: at most `entries_`
    // requests are ever outstanding, so the completion ring cannot overflow) has no room
    bool enqueue(Req *r)
    {
        const uint32_t tail = sq_tail_->load(std::memory_order_relaxed);
        if (inflight_ >= entries_ || tail - sq_head_->load(std::memory_order_acquire) >= entries_)
            return false;
        struct io_uring_sqe *sqe = &sqes_[tail & sq_mask_];
        memset(sqe, 0, sizeof(*sqe));
        sqe->opcode = IORING_OP_READ;
        sqe->fd = r->fd;
        sqe->addr = (uint64_t)(uintptr_t)r->dst;
        sqe->len = (uint32_t)r->len;
        sqe->off = r->off;
        sqe->user_data = (uint64_t)(uintptr_t)r;
        sq_array_[tail & sq_mask_] = tail & sq_mask_;
        sq_tail_->store(tail + 1, std::memory_order_release);
        unsubmitted_++;
        inflight_++;
        r->live_at = live_.size();
        live_.push_back(r);
        return true;
    }

    // hand the queued entries to the kernel; wait_for > 0: also sleep until that many completions are there
    bool enter(unsigned wait_for)
    {
        if (!unsubmitted_ && !wait_for)
            return true;
        for (;;) {
            const long rc = syscall(__NR_io_uring_enter, fd_, unsubmitted_, wait_for, wait_for ? IORING_ENTER_GETEVENTS : 0u,
                                    nullptr, 0);
            if (rc >= 0) {
                unsubmitted_ -= (unsigned)rc <= unsubmitted_ ? (unsigned)rc : unsubmitted_;
                return true;
            }
            if (errno == EINTR)
                continue;
            if (errno == EAGAIN || errno == EBUSY) {  // the kernel wants completions taken first
                if (!reap())
                    usleep(50);
                if (wait_for)
                    return true;  // (the caller looks at its batch again)
                continue;
            }
            return false;
        }
    }

    static int finish_buffered(Req *r, uint64_t done)
    {
        while (done < r->len) {
            const ssize_t got = pread(r->fd_buffered, r->dst + done, r->len - done, (off_t)(r->off + done));
            if (got < 0 && errno == EINTR)
                continue;
            if (got <= 0)
                return got < 0 ? errno : EIO;
            done += (uint64_t)got;
        }
        return 0;
    }

    void complete(Req *r, int res)
    {
        int err = 0;
        if (res >= 0 && (uint64_t)res < r->len)
            err = finish_buffered(r, (uint64_t)res);  // short: the unaligned end of the file behind an O_DIRECT read
        else if (res == -EINVAL || res == -EAGAIN || res == -EINTR || res == -EOPNOTSUPP)
            err = finish_buffered(r, 0);              // the direct descriptor refuses this request after all
        else if (res < 0)
            err = -res;
        if (err && !r->batch->error)
            r->batch->error = err;
        r->batch->pending--;
        delete r;
    }

    // take every completion that is there (and refill the ring from the backlog); returns how many
    unsigned reap()
    {
        unsigned n = 0;
        uint32_t head = cq_head_->load(std::memory_order_relaxed);
        while (head != cq_tail_->load(std::memory_order_acquire)) {
            const struct io_uring_cqe c = cqes_[head & cq_mask_];
            head++;
            cq_head_->store(head, std::memory_order_release);
            inflight_--;
            Req *r = (Req *)(uintptr_t)c.user_data;
            if (r->live_at < live_.size() && live_[r->live_at] == r) {  // (swap-remove from the in-flight set)
                live_[r->live_at] = live_.back();
                live_[r->live_at]->live_at = r->live_at;
                live_.pop_back();
            }
            complete(r, c.res);
            n++;
        }
        while (!dead_ && !backlog_.empty() && enqueue(backlog_.front()))
            backlog_.pop_front();
        return n;
    }

    int fd_ = -1;
    unsigned entries_ = 0, unsubmitted_ = 0, inflight_ = 0;
    void *sq_ptr_ = nullptr, *cq_ptr_ = nullptr;
    size_t sq_bytes_ = 0, cq_bytes_ = 0, sqe_bytes_ = 0;
    struct io_uring_sqe *sqes_ = nullptr;
    std::atomic<uint32_t> *sq_head_ = nullptr, *sq_tail_ = nullptr, *cq_head_ = nullptr, *cq_tail_ = nullptr;
    uint32_t sq_mask_ = 0, cq_mask_ = 0;
    uint32_t *sq_array_ = nullptr;
    struct io_uring_cqe *cqes_ = nullptr;
    std::deque<Req *> backlog_;
    std::vector<Req *> live_;  // what the kernel has
    ReadBatch orphans_{};      // where completions of a given-up ring's stragglers go
    bool dead_ = false;
    bool poisoned_ = false;
};

}  // namespace papr_rt
#endif
