// rsbwt_gpubwt.hpp -- header-only C++ shim: ReadServer's `class BWT` served by the HIP engine.
//
// Meant to be compiled INSIDE the ReadServer tree (it includes the reference's own
// include/bwt/bwt.h and include/bwt/query.h) and linked with librsbwt.so.  With it the reference's
// callers keep their shape:
//
//     std::unique_ptr<BWT> pbwt(new GpuBWT(prefix + ".bwt"));      // was: new RLEBWT(...)
//     BWTInterval itv = findInterval(pbwt.get(), query);             // src/bwt/query.cpp unchanged
//
// Every virtual of include/bwt/bwt.h:6-15 forwards to the scalar mirror of include/rsbwt.h.  Each
// such call is one GPU round trip: correct, but the point of the engine is the batched entry
// below, which a micro-batching service loop should use instead.
#ifndef RSBWT_GPUBWT_HPP
#define RSBWT_GPUBWT_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "bwt.h"    // reference: include/bwt/bwt.h
#include "query.h"  // reference: include/bwt/query.h (BWTInterval)
#include "rsbwt.h"

class GpuBWT : public BWT {
 public:
  explicit GpuBWT(const std::string& filename, int device = 0) : h_(nullptr) {
    if (rsbwt_open(filename.c_str(), device, 0u, &h_) != RSBWT_OK)
      throw std::runtime_error(rsbwt_last_error());
  }
  ~GpuBWT() { rsbwt_close(h_); }  // NB: class BWT has no virtual destructor (bwt.h:6-15)
  GpuBWT(const GpuBWT&) = delete;
  GpuBWT& operator=(const GpuBWT&) = delete;

  char getChar(const uint64_t& index) const {
    char c = '$';
    check(rsbwt_char(h_, index, &c));
    return c;
  }
  uint64_t getOccAt(const char& b, const uint64_t& bc) const {
    uint64_t v = 0;
    check(rsbwt_occ_at(h_, b, bc, &v));
    return v;
  }
  uint64_t getOcc(const char& b, const uint64_t& index) const {
    uint64_t v = 0;
    check(rsbwt_occ(h_, b, index, &v));
    return v;
  }
  char getF(const uint64_t& index) const { return rsbwt_f(h_, index); }
  uint64_t getPC(const char& b) const { return rsbwt_pc(h_, b); }
  uint64_t getBWLen() const { return rsbwt_bwlen(h_); }

  rsbwt_t* handle() const { return h_; }

 private:
  static void check(int rc) {
    if (rc != RSBWT_OK) throw std::runtime_error(rsbwt_last_error());
  }
  rsbwt_t* h_;
};

// Batched findInterval (src/bwt/query.cpp:24-41) for k-mers of one length.
inline std::vector<BWTInterval> findIntervals(const GpuBWT* pBWT, const std::vector<std::string>& ws) {
  std::vector<BWTInterval> out(ws.size());
  if (ws.empty()) return out;
  const size_t k = ws[0].size();
  std::string flat;
  flat.reserve(ws.size() * k);
  for (const std::string& w : ws) {
    if (w.size() != k) throw std::invalid_argument("findIntervals: k-mers of one batch must have one length");
    flat += w;
  }
  std::vector<uint64_t> lo(ws.size()), up(ws.size());
  if (rsbwt_find_intervals(pBWT->handle(), flat.data(), ws.size(), (uint32_t)k, k ? k : 1, lo.data(),
                           up.data()) != RSBWT_OK)
    throw std::runtime_error(rsbwt_last_error());
  for (size_t i = 0; i < ws.size(); ++i) {
    out[i].lower = lo[i];
    out[i].upper = up[i];
  }
  return out;
}

#endif
