// include/circl/serving.hpp -- kem.Scheme for callers that cannot batch, on the ASYNCHRONOUS table API (include/circl_hip.h:
// circl_hip_keytable_async_start, *_table_submit, circl_hip_poll / circl_hip_wait).  The compiled counterpart of
// go/kem/mlkem/hipbatch/{reactor,serving}.go (no Go toolchain exists on any box this repository can use; this header is built and run
// by tests/host_mirror_test.cpp on the GPU):
//
//   caller thread / event loop   hands {input, promise} to its key's reactor and gets a std::future (EncapsulateAsync / DecapsulateAsync;
//                                the blocking forms wait on it).  It never enters the library.
//   reactor thread               ONE per key object (= per device queue): takes whatever is pending, copies the inputs side by side and
//                                submits them with ONE call (n items, one ticket; the call returns at once), polls the head of its ticket
//                                ring, fulfils the promises of finished calls.  With nothing pending it blocks in circl_hip_wait on the
//                                oldest ticket -- the one thread per device a host needs to block.
//
// Reference shape being served: kem.Scheme.Encapsulate / Decapsulate with one key and one item per call from whichever goroutine owns the
// connection (kem/mlkem/mlkem768/kyber.go:347-386, hpke/algs.go:283-285, kem/hybrid/hybrid.go:95-99).  Results are those of kem::Scheme.
#pragma once
#include <condition_variable>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>

#include "kem.hpp"

namespace circl {
namespace kem {

// One parsed key with a resident table, an asynchronous queue and its reactor.
class ServingKey {
  public:
    using Result = std::pair<Bytes, Bytes>;  // {ct, ss}; ct is empty for a decapsulation
    ServingKey(int param, bool private_key, const Bytes &packed, int device, size_t max_items, size_t window)
        : private_(private_key), window_(window), call_max_(max_items / 4 ? max_items / 4 : 1), ct_size_(circl_hip_mlkem_ct_size(param)),
          in_size_(private_key ? ct_size_ : 32) {
        uint8_t verdict = 0;
        check(circl_hip_mlkem_keytable_new(param, private_key ? 1 : 0, packed.data(), 1, device, &verdict, &table_));
        if (verdict == CIRCL_HIP_ITEM_ERR_PRIVKEY) {
            circl_hip_keytable_free(table_);
            throw ErrPrivKey();
        }
        const int rc = circl_hip_keytable_async_start(table_, max_items, 0, 0);
        if (rc != CIRCL_HIP_OK) {
            circl_hip_keytable_free(table_);
            check(rc);
        }
        arena_ct_.resize(private_ ? 0 : window_ * ct_size_);
        arena_ss_.resize(window_ * 32);
        arena_st_.resize(window_);
        scratch_.resize(call_max_ * in_size_);
        reactor_ = std::thread([this] { loop(); });
    }
    ~ServingKey() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        reactor_.join();  // (finishes what is in flight first)
        while (circl_hip_keytable_close(table_) == CIRCL_HIP_EBUSY) std::this_thread::yield();  // nobody is inside any more: the reactor was the only caller
    }
    ServingKey(const ServingKey &) = delete;
    ServingKey &operator=(const ServingKey &) = delete;

    // hands the request to the reactor; the future carries the result or the exception kem::Scheme would throw
    std::future<Result> Submit(Bytes in) {
        if (in.size() != in_size_) {
            if (private_) throw ErrCiphertextSize();
            throw ErrSeedSize();
        }
        Request rq;
        rq.in = std::move(in);
        std::future<Result> f = rq.done.get_future();
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (quit_) throw ErrDevice("the key object is being closed");
            pending_.push_back(std::move(rq));
        }
        cv_.notify_one();
        return f;
    }
    bool IsPrivate() const { return private_; }

  private:
    struct Request {
        Bytes in;
        std::promise<Result> done;
    };
    struct Flight {  // one submitted call: requests in arena slots [slot, slot + reqs.size())
        uint64_t ticket;
        size_t slot;
        std::vector<Request> reqs;
    };
    static void check(int rc) {
        if (rc != CIRCL_HIP_OK) throw ErrDevice(std::string("error ") + std::to_string(rc) + " " + circl_hip_last_error());
    }
    void loop() {
        std::deque<Flight> fifo;
        size_t tail = 0, in_use = 0;
        std::vector<Request> batch;
        for (;;) {
            // ---- take what is pending (sleep only while nothing is pending AND nothing is in flight) ----
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (fifo.empty()) cv_.wait(lk, [&] { return quit_ || !pending_.empty(); });
                while (!pending_.empty() && batch.size() < window_) {
                    batch.push_back(std::move(pending_.front()));
                    pending_.pop_front();
                }
                if (quit_ && batch.empty() && fifo.empty()) return;
            }
            // ---- submit: as many requests per call as a call may hold and the arena ring has contiguous room for ----
            size_t done = 0;
            bool eagain = false;
            while (done < batch.size() && in_use < window_) {
                const size_t n = std::min({batch.size() - done, call_max_, window_ - in_use, window_ - tail});
                for (size_t i = 0; i < n; i++) std::copy(batch[done + i].in.begin(), batch[done + i].in.end(), scratch_.begin() + i * in_size_);
                uint64_t ticket = 0;
                const int rc = private_ ? circl_hip_mlkem_decaps_table_submit(table_, nullptr, scratch_.data(), &arena_ss_[tail * 32], &arena_st_[tail], n, &ticket)
                                        : circl_hip_mlkem_encaps_table_submit(table_, nullptr, scratch_.data(), &arena_ct_[tail * ct_size_], &arena_ss_[tail * 32],
                                                                              &arena_st_[tail], n, &ticket);
                std::fill(scratch_.begin(), scratch_.begin() + n * in_size_, (uint8_t)0);  // (the library has its copy; encapsulation seeds are secret)
                if (rc == CIRCL_HIP_EAGAIN) { eagain = true; break; }  // every device batch busy: a completion has to come first
                Flight f{ticket, tail, {}};
                for (size_t i = 0; i < n; i++) f.reqs.push_back(std::move(batch[done + i]));
                if (rc != CIRCL_HIP_OK) {
                    for (auto &rq : f.reqs) rq.done.set_exception(std::make_exception_ptr(ErrDevice(std::string("submit: error ") + std::to_string(rc))));
                } else {
                    fifo.push_back(std::move(f));
                    tail = (tail + n) % window_;
                    in_use += n;
                }
                done += n;
            }
            batch.erase(batch.begin(), batch.begin() + done);
            // ---- reap the head of the ring; with nothing else to do, block on it (briefly: new requests must not wait long) ----
            bool reaped = false;
            while (!fifo.empty()) {
                int8_t state = 0;
                circl_hip_poll(table_, &fifo.front().ticket, 1, &state);
                if (state == 0) {
                    if (reaped || (!batch.empty() && !eagain && in_use < window_)) break;  // (more could be submitted right now: do that first)
                    bool idle;
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        idle = pending_.empty() || eagain || in_use == window_;
                    }
                    if (!idle) break;
                    state = (int8_t)circl_hip_wait(table_, fifo.front().ticket, 50);
                    if (state == 0) break;
                }
                Flight &f = fifo.front();
                for (size_t i = 0; i < f.reqs.size(); i++) {
                    const size_t s = f.slot + i;
                    if (state != 1) {
                        f.reqs[i].done.set_exception(std::make_exception_ptr(ErrDevice(std::string("batch failed: ") + std::to_string((int)state))));
                    } else if (arena_st_[s] == CIRCL_HIP_ITEM_ERR_PUBKEY) {
                        f.reqs[i].done.set_exception(std::make_exception_ptr(ErrPubKey()));
                    } else if (arena_st_[s] == CIRCL_HIP_ITEM_ERR_PRIVKEY) {
                        f.reqs[i].done.set_exception(std::make_exception_ptr(ErrPrivKey()));
                    } else {
                        Result r;
                        if (!private_) r.first.assign(arena_ct_.begin() + s * ct_size_, arena_ct_.begin() + (s + 1) * ct_size_);
                        r.second.assign(arena_ss_.begin() + s * 32, arena_ss_.begin() + (s + 1) * 32);
                        f.reqs[i].done.set_value(std::move(r));
                    }
                    std::fill(arena_ss_.begin() + s * 32, arena_ss_.begin() + (s + 1) * 32, (uint8_t)0);  // the arena's copy of the shared secret
                }
                in_use -= f.reqs.size();
                fifo.pop_front();
                reaped = true;
            }
        }
    }

    const bool private_;
    const size_t window_, call_max_, ct_size_, in_size_;
    circl_hip_keytable *table_ = nullptr;
    Bytes arena_ct_, arena_ss_, arena_st_, scratch_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Request> pending_;
    bool quit_ = false;
    std::thread reactor_;
};

// The serving form of an ML-KEM scheme: key objects carry a reactor; single operations are futures.
class Serving {
  public:
    Serving(const Scheme *scheme, int device, size_t max_items = 2048, size_t window = 4096)
        : scheme_(scheme), param_(scheme->PublicKeySize() == 800 ? 512 : scheme->PublicKeySize() == 1184 ? 768 : 1024), device_(device), max_items_(max_items),
          window_(window) {
        if (device < 0) throw ErrDevice("a serving scheme lives on ONE device (a reactor owns one queue)");
    }
    std::shared_ptr<ServingKey> UnmarshalBinaryPublicKey(const Bytes &buf) const {
        (void)scheme_->UnmarshalBinaryPublicKey(buf);  // the reference's checks and errors (size, canonical encoding)
        return std::make_shared<ServingKey>(param_, false, buf, device_, max_items_, window_);
    }
    std::shared_ptr<ServingKey> UnmarshalBinaryPrivateKey(const Bytes &buf) const {
        if ((int)buf.size() != scheme_->PrivateKeySize()) throw ErrPrivKeySize();
        return std::make_shared<ServingKey>(param_, true, buf, device_, max_items_, window_);  // throws ErrPrivKey on a stored-hash mismatch
    }
    // one row of whatever launch the key's reactor submits next
    std::future<ServingKey::Result> EncapsulateAsync(ServingKey &pk, const Bytes &seed) const {
        if (pk.IsPrivate()) throw ErrTypeMismatch();
        return pk.Submit(seed);
    }
    std::future<ServingKey::Result> DecapsulateAsync(ServingKey &sk, const Bytes &ct) const {
        if (!sk.IsPrivate()) throw ErrTypeMismatch();
        return sk.Submit(ct);
    }
    ServingKey::Result EncapsulateDeterministically(ServingKey &pk, const Bytes &seed) const { return EncapsulateAsync(pk, seed).get(); }
    Bytes Decapsulate(ServingKey &sk, const Bytes &ct) const { return DecapsulateAsync(sk, ct).get().second; }

  private:
    const Scheme *scheme_;
    int param_, device_;
    size_t max_items_, window_;
};

}  // namespace kem
}  // namespace circl
