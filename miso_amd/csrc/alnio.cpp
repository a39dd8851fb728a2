// alnio.cpp -- BAM / SAM reader behind include/miso_alnio.h.
//
// The reference reaches alignments through pysam, one Python object per read
// (misopy/sam_utils.py:139-186, 207-442).  Here a file is decoded once into columns:
//   BAM : mmap, walk the BGZF block headers (BSIZE), inflate all blocks in parallel straight into
//         one buffer (each block knows its ISIZE), then one serial pass over the records
//         (SAM spec v1, section 4.2);
//   SAM : one pass over the text.
// and indexed in memory by (reference, position) with a prefix maximum of the end coordinates, so
// an event's fetch is a binary search + a short scan (no .bai needed, unsorted input accepted).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "miso_alnio.h"
#include "miso_amd.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

const char kCigarOps[] = "MIDNSHP=X";

}  // namespace

struct miso_alnfile {
  bool is_bam = false;
  std::vector<std::string> ref_names;
  std::vector<int64_t> ref_len;
  std::unordered_map<std::string, int> ref_index;
  // columns, file order
  std::vector<int32_t> ref_id, pos, end, flag, l_seq;
  std::vector<uint64_t> cigar_off{0}, name_off{0};
  std::vector<uint32_t> cigar;
  std::vector<char> names;
  // index: records with a reference, ordered by (ref, pos, file order)
  std::vector<int64_t> order;
  std::vector<int64_t> ref_begin;   // n_refs + 1 offsets into `order`
  std::vector<int32_t> pmax_end;    // running maximum of `end` inside each reference's segment

  int64_t n() const { return static_cast<int64_t>(pos.size()); }

  void push(int32_t rid, int32_t p, int32_t fl, int32_t lseq, const uint32_t *cg, size_t ncg,
            const char *name, size_t lname) {
    int64_t reflen = 0;
    for (size_t i = 0; i < ncg; i++) {
      const uint32_t op = cg[i] & 15u;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += cg[i] >> 4;
    }
    // htslib bam_endpos: unmapped or no reference-consuming op -> pos + 1
    const int32_t e = ((fl & 4) || reflen == 0) ? p + 1 : static_cast<int32_t>(p + reflen);
    ref_id.push_back(rid); pos.push_back(p); end.push_back(e); flag.push_back(fl);
    l_seq.push_back(lseq);
    cigar.insert(cigar.end(), cg, cg + ncg);
    cigar_off.push_back(cigar.size());
    names.insert(names.end(), name, name + lname);
    name_off.push_back(names.size());
  }

  // (reference, position, file order): records are dealt to their reference's range first (a counting pass, stable),
  // then every reference's range is ordered by position on its own thread -- a coordinate-sorted BAM (the usual input)
  // is recognised and left as it is; 20 M unsorted records: 1.09 s single-threaded before, see profiles/r03_e2e.txt.
  void build_index() {
    const int nref = static_cast<int>(ref_names.size());
    ref_begin.assign(nref + 1, 0);
    for (int64_t i = 0; i < n(); i++)
      if (ref_id[i] >= 0 && ref_id[i] < nref) ref_begin[ref_id[i] + 1]++;
    for (int r = 0; r < nref; r++) ref_begin[r + 1] += ref_begin[r];
    order.assign(static_cast<size_t>(ref_begin[nref]), 0);
    {
      std::vector<int64_t> fill(ref_begin.begin(), ref_begin.end() - 1);
      for (int64_t i = 0; i < n(); i++)
        if (ref_id[i] >= 0 && ref_id[i] < nref) order[static_cast<size_t>(fill[ref_id[i]]++)] = i;
    }
    pmax_end.resize(order.size());
    std::atomic<int> next{0};
    auto work = [&] {
      for (;;) {
        const int r = next.fetch_add(1);
        if (r >= nref) return;
        const auto lo = order.begin() + ref_begin[r], hi = order.begin() + ref_begin[r + 1];
        bool sorted = true;
        for (auto it = lo; sorted && it != hi && it + 1 != hi; ++it) sorted = pos[*it] <= pos[*(it + 1)];
        if (!sorted) std::stable_sort(lo, hi, [&](int64_t a, int64_t b) { return pos[a] < pos[b]; });
        int32_t m = INT32_MIN;
        for (int64_t j = ref_begin[r]; j < ref_begin[r + 1]; j++) {
          m = std::max(m, end[order[static_cast<size_t>(j)]]);
          pmax_end[static_cast<size_t>(j)] = m;
        }
      }
    };
    const int T = std::max(1, std::min(nref, miso_usable_threads()));
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
  }

  template <class F> void for_overlaps(int ref, int64_t start, int64_t stop, F &&f) const {
    if (ref < 0 || ref + 1 >= static_cast<int>(ref_begin.size())) return;
    const int64_t lo = ref_begin[ref], hi = ref_begin[ref + 1];
    // first record whose running max end exceeds `start`: nothing before it can overlap
    const int64_t first = std::upper_bound(pmax_end.begin() + lo, pmax_end.begin() + hi, start,
                                           [](int64_t s, int32_t e) { return s < e; }) -
                          pmax_end.begin();
    for (int64_t j = first; j < hi; j++) {
      const int64_t i = order[j];
      if (pos[i] >= stop) break;
      if (end[i] > start) f(i);
    }
  }
};

namespace {

struct Mapped {
  const unsigned char *p = nullptr;
  size_t len = 0;
  int fd = -1;
  ~Mapped() {
    if (p && len) munmap(const_cast<unsigned char *>(p), len);
    if (fd >= 0) close(fd);
  }
};

inline uint32_t rd32(const unsigned char *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const unsigned char *p) { uint16_t v; std::memcpy(&v, p, 2); return v; }

struct Block { size_t in_off, in_len, out_off, out_len; uint32_t crc; };

int inflate_all(const Mapped &m, int n_threads, std::vector<unsigned char> &out) {
  std::vector<Block> blocks;
  size_t off = 0, total = 0;
  while (off < m.len) {
    if (m.len - off < 18 || m.p[off] != 31 || m.p[off + 1] != 139 || m.p[off + 2] != 8 ||
        !(m.p[off + 3] & 4))
      return fail(MISO_EINVAL, "not a BGZF block at offset " + std::to_string(off));
    const size_t xlen = rd16(m.p + off + 10);
    size_t x = off + 12, bsize = 0;
    const size_t xend = x + xlen;
    if (xend > m.len) return fail(MISO_EINVAL, "truncated BGZF header");
    while (x + 4 <= xend) {
      const size_t slen = rd16(m.p + x + 2);
      if (m.p[x] == 'B' && m.p[x + 1] == 'C' && slen == 2) bsize = static_cast<size_t>(rd16(m.p + x + 4)) + 1;
      x += 4 + slen;
    }
    if (bsize < xlen + 20 || off + bsize > m.len) return fail(MISO_EINVAL, "bad BGZF block size");
    const size_t isize = rd32(m.p + off + bsize - 4);
    blocks.push_back({off + 12 + xlen, bsize - xlen - 20, total, isize, rd32(m.p + off + bsize - 8)});
    total += isize;
    off += bsize;
  }
  try { out.resize(total); } catch (...) { return fail(MISO_ENOMEM, "out of memory inflating the BAM file"); }
  std::atomic<size_t> next{0};
  std::atomic<int> bad{0};
  auto work = [&] {
    z_stream zs;
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= blocks.size() || bad.load()) return;
      const Block &k = blocks[b];
      if (k.out_len == 0) continue;
      std::memset(&zs, 0, sizeof zs);
      if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
      zs.next_in = const_cast<unsigned char *>(m.p + k.in_off);
      zs.avail_in = static_cast<uInt>(k.in_len);
      zs.next_out = out.data() + k.out_off;
      zs.avail_out = static_cast<uInt>(k.out_len);
      const int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = 1; return; }
      // the block's CRC32 sits before ISIZE: a flipped bit that still inflates must not pass
      if (crc32(crc32(0L, Z_NULL, 0), out.data() + k.out_off, static_cast<uInt>(k.out_len)) != k.crc) { bad = 1; return; }
    }
  };
  const int T = std::max(1, std::min<int>(n_threads, static_cast<int>(blocks.size())));
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
  if (bad.load()) return fail(MISO_EINVAL, "corrupt BGZF block (inflate or CRC32 failed)");
  return 0;
}

int parse_bam(const std::vector<unsigned char> &d, miso_alnfile &f, int n_threads) {
  const size_t n = d.size();
  size_t o = 0;
  auto need = [&](size_t k) { return o + k <= n; };
  if (!need(12) || std::memcmp(d.data(), "BAM\1", 4) != 0) return fail(MISO_EINVAL, "missing BAM magic");
  const uint32_t l_text = rd32(d.data() + 4);
  o = 8 + static_cast<size_t>(l_text);
  if (!need(4)) return fail(MISO_EINVAL, "truncated BAM header");
  const uint32_t n_ref = rd32(d.data() + o); o += 4;
  for (uint32_t r = 0; r < n_ref; r++) {
    if (!need(4)) return fail(MISO_EINVAL, "truncated BAM reference list");
    const uint32_t l_name = rd32(d.data() + o); o += 4;
    if (!need(static_cast<size_t>(l_name) + 4) || l_name == 0) return fail(MISO_EINVAL, "truncated BAM reference list");
    std::string name(reinterpret_cast<const char *>(d.data() + o), l_name - 1);
    o += l_name;
    f.ref_len.push_back(static_cast<int32_t>(rd32(d.data() + o))); o += 4;
    f.ref_index.emplace(name, static_cast<int>(f.ref_names.size()));
    f.ref_names.push_back(std::move(name));
  }
  // pass 1 (serial, a few ns per record): where every record starts, and the running totals of CIGAR
  // operations and name bytes, so that pass 2 can fill preallocated columns from all threads
  std::vector<size_t> rec;
  while (o < n) {
    if (!need(4)) return fail(MISO_EINVAL, "truncated BAM record");
    const uint32_t bs = rd32(d.data() + o);
    if (bs < 32 || o + 4 + static_cast<size_t>(bs) > n) return fail(MISO_EINVAL, "truncated BAM record");
    const unsigned char *r = d.data() + o + 4;
    const uint32_t l_name = r[8], n_cig = rd16(r + 12);
    if (32 + static_cast<size_t>(l_name) + 4 * static_cast<size_t>(n_cig) > bs || l_name == 0)
      return fail(MISO_EINVAL, "malformed BAM record");
    rec.push_back(o + 4);
    f.cigar_off.push_back(f.cigar_off.back() + n_cig);
    f.name_off.push_back(f.name_off.back() + (l_name - 1));
    o += 4 + static_cast<size_t>(bs);
  }
  const size_t N = rec.size();
  f.ref_id.resize(N); f.pos.resize(N); f.end.resize(N); f.flag.resize(N); f.l_seq.resize(N);
  f.cigar.resize(f.cigar_off.back());
  f.names.resize(f.name_off.back());
  auto fill = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      const unsigned char *r = d.data() + rec[i];
      const int32_t p = static_cast<int32_t>(rd32(r + 4));
      const uint32_t l_name = r[8], n_cig = rd16(r + 12), fl = rd16(r + 14);
      uint32_t *cg = f.cigar.data() + f.cigar_off[i];
      if (n_cig) std::memcpy(cg, r + 32 + l_name, 4 * static_cast<size_t>(n_cig));   // unaligned in the record
      int64_t reflen = 0;
      for (uint32_t c = 0; c < n_cig; c++) {
        const uint32_t op = cg[c] & 15u;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += cg[c] >> 4;
      }
      f.ref_id[i] = static_cast<int32_t>(rd32(r));
      f.pos[i] = p;
      f.end[i] = ((fl & 4) || reflen == 0) ? p + 1 : static_cast<int32_t>(p + reflen);   // bam_endpos
      f.flag[i] = static_cast<int32_t>(fl);
      f.l_seq[i] = static_cast<int32_t>(rd32(r + 16));
      std::memcpy(f.names.data() + f.name_off[i], r + 32, l_name - 1);
    }
  };
  const size_t T = std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(n_threads), N / 65536 + 1));
  std::vector<std::thread> th;
  for (size_t t = 1; t < T; t++) th.emplace_back(fill, N * t / T, N * (t + 1) / T);
  fill(0, N / T);
  for (auto &t : th) t.join();
  return 0;
}

// strict: a reference name missing from the header is an error for this chunk (the caller then falls
// back to one serial pass, which numbers header-less references in order of appearance)
int parse_sam(const unsigned char *p, size_t len, miso_alnfile &f, bool strict = false) {
  const char *s = reinterpret_cast<const char *>(p), *e = s + len;
  std::vector<uint32_t> cg;
  size_t lineno = 0;
  while (s < e) {
    const char *nl = static_cast<const char *>(std::memchr(s, '\n', e - s));
    const char *le = nl ? nl : e;
    const char *next = nl ? nl + 1 : e;
    if (le > s && le[-1] == '\r') le--;
    lineno++;
    if (le == s) { s = next; continue; }
    if (*s == '@') {
      if (le - s >= 3 && s[1] == 'S' && s[2] == 'Q') {
        std::string name; int64_t ln = 0;
        const char *q = s;
        while (q < le) {
          const char *t = static_cast<const char *>(std::memchr(q, '\t', le - q));
          const char *fe = t ? t : le;
          if (fe - q > 3 && q[0] == 'S' && q[1] == 'N' && q[2] == ':') name.assign(q + 3, fe);
          if (fe - q > 3 && q[0] == 'L' && q[1] == 'N' && q[2] == ':') ln = std::strtoll(std::string(q + 3, fe).c_str(), nullptr, 10);
          q = t ? t + 1 : le;
        }
        if (!name.empty() && !f.ref_index.count(name)) {
          f.ref_index.emplace(name, static_cast<int>(f.ref_names.size()));
          f.ref_names.push_back(name);
          f.ref_len.push_back(ln);
        }
      }
      s = next;
      continue;
    }
    const char *fld[11]; size_t fl_len[11]; int nf = 0;
    const char *q = s;
    while (nf < 11) {
      const char *t = static_cast<const char *>(std::memchr(q, '\t', le - q));
      fld[nf] = q; fl_len[nf] = (t ? t : le) - q; nf++;
      if (!t) break;
      q = t + 1;
    }
    if (nf < 11) return fail(MISO_EINVAL, "SAM line " + std::to_string(lineno) + ": fewer than 11 fields");
    const int32_t flag = static_cast<int32_t>(std::strtol(std::string(fld[1], fl_len[1]).c_str(), nullptr, 10));
    const std::string rname(fld[2], fl_len[2]);
    int32_t rid = -1;
    if (rname != "*") {
      auto it = f.ref_index.find(rname);
      if (it == f.ref_index.end() && strict) return fail(MISO_FAILURE, "reference not in the header");
      if (it == f.ref_index.end()) {  // header-less SAM: references in order of appearance
        rid = static_cast<int32_t>(f.ref_names.size());
        f.ref_index.emplace(rname, rid);
        f.ref_names.push_back(rname);
        f.ref_len.push_back(0);
      } else {
        rid = it->second;
      }
    }
    const int32_t pos1 = static_cast<int32_t>(std::strtol(std::string(fld[3], fl_len[3]).c_str(), nullptr, 10));
    cg.clear();
    if (!(fl_len[5] == 1 && fld[5][0] == '*')) {
      const char *c = fld[5], *ce = c + fl_len[5];
      while (c < ce) {
        uint64_t v = 0; bool digits = false;
        while (c < ce && *c >= '0' && *c <= '9') { v = v * 10 + (*c - '0'); c++; digits = true; }
        const char *opp = (c < ce) ? std::strchr(kCigarOps, *c) : nullptr;
        if (!digits || !opp || *c == '\0' || v >= (1u << 28))
          return fail(MISO_EINVAL, "SAM line " + std::to_string(lineno) + ": bad CIGAR");
        cg.push_back(static_cast<uint32_t>(v << 4) | static_cast<uint32_t>(opp - kCigarOps));
        c++;
      }
    }
    const int32_t l_seq = (fl_len[9] == 1 && fld[9][0] == '*') ? 0 : static_cast<int32_t>(fl_len[9]);
    f.push(rid, pos1 - 1, flag, l_seq, cg.data(), cg.size(), fld[0], fl_len[0]);
    s = next;
  }
  return 0;
}


// SAM text on all cores: the header serially, then the records in chunks cut at line ends, each into
// its own columns, concatenated in file order.
int parse_sam_parallel(const unsigned char *p, size_t len, miso_alnfile &f, int n_threads) {
  const char *s = reinterpret_cast<const char *>(p), *e = s + len;
  const char *body = s;
  while (body < e && *body == '@') {
    const char *nl = static_cast<const char *>(std::memchr(body, '\n', e - body));
    body = nl ? nl + 1 : e;
  }
  const size_t body_len = static_cast<size_t>(e - body);
  const int T = static_cast<int>(std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(n_threads), body_len >> 22)));
  if (T <= 1) return parse_sam(p, len, f);
  int rc = parse_sam(p, static_cast<size_t>(body - s), f);   // header only
  if (rc) return rc;
  std::vector<const char *> cut(T + 1, e);
  cut[0] = body;
  for (int t = 1; t < T; t++) {
    const char *q = body + body_len * t / T;
    const char *nl = static_cast<const char *>(std::memchr(q, '\n', e - q));
    cut[t] = nl ? nl + 1 : e;
  }
  std::vector<miso_alnfile> part(T);
  std::vector<int> prc(T, 0);
  std::vector<std::string> perr(T);
  auto work = [&](int t) {
    part[t].ref_names = f.ref_names; part[t].ref_len = f.ref_len; part[t].ref_index = f.ref_index;
    prc[t] = parse_sam(reinterpret_cast<const unsigned char *>(cut[t]), static_cast<size_t>(cut[t + 1] - cut[t]),
                       part[t], true);
    if (prc[t]) perr[t] = g_err;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back(work, t);
  work(0);
  for (auto &x : th) x.join();
  for (int t = 0; t < T; t++) {
    if (prc[t] == MISO_FAILURE) {   // a reference the header does not list: serial pass
      f = miso_alnfile();
      return parse_sam(p, len, f);
    }
    if (prc[t]) return fail(prc[t], perr[t] + " (line numbers count from the start of a " +
                                    std::to_string(T) + "-way split)");
  }
  size_t n = 0, nc = 0, nn = 0;
  for (auto &q : part) { n += q.pos.size(); nc += q.cigar.size(); nn += q.names.size(); }
  f.ref_id.reserve(n); f.pos.reserve(n); f.end.reserve(n); f.flag.reserve(n); f.l_seq.reserve(n);
  f.cigar.reserve(nc); f.names.reserve(nn); f.cigar_off.reserve(n + 1); f.name_off.reserve(n + 1);
  for (auto &q : part) {
    f.ref_id.insert(f.ref_id.end(), q.ref_id.begin(), q.ref_id.end());
    f.pos.insert(f.pos.end(), q.pos.begin(), q.pos.end());
    f.end.insert(f.end.end(), q.end.begin(), q.end.end());
    f.flag.insert(f.flag.end(), q.flag.begin(), q.flag.end());
    f.l_seq.insert(f.l_seq.end(), q.l_seq.begin(), q.l_seq.end());
    const uint64_t c0 = f.cigar.size(), n0 = f.names.size();
    f.cigar.insert(f.cigar.end(), q.cigar.begin(), q.cigar.end());
    f.names.insert(f.names.end(), q.names.begin(), q.names.end());
    for (size_t i = 1; i < q.cigar_off.size(); i++) f.cigar_off.push_back(c0 + q.cigar_off[i]);
    for (size_t i = 1; i < q.name_off.size(); i++) f.name_off.push_back(n0 + q.name_off[i]);
    q = miso_alnfile();
  }
  return 0;
}
}  // namespace

// host threads this process may really use: affinity mask capped by the cgroup CPU quota
extern "C" int miso_usable_threads(void) {
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<long>(n, CPU_COUNT(&set));
  if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32]; long period = 0;
    if (std::fscanf(fp, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0)
      n = std::min<long>(n, std::max<long>(1, std::atol(quota) / period));
    std::fclose(fp);
  }
  return static_cast<int>(std::max<long>(1, std::min<long>(n, 64)));
}

namespace {

int usable_threads() { return miso_usable_threads(); }

void append_cigar_string(const miso_alnfile &f, int64_t i, std::string &out) {
  char buf[16];
  for (uint64_t c = f.cigar_off[i]; c < f.cigar_off[i + 1]; c++) {
    uint32_t v = f.cigar[c] >> 4;
    const uint32_t op = f.cigar[c] & 15u;
    int k = 0;
    do { buf[k++] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
    while (k) out.push_back(buf[--k]);
    out.push_back(op < 9 ? kCigarOps[op] : '?');
  }
  out.push_back('\0');
}

// misopy/sam_utils.py:195-204: endswith /1 /2 #1 #2 -> drop the last THREE characters (sic)
std::string_view strip_mate_id(std::string_view name) {
  const size_t n = name.size();
  if (n >= 2 && (name[n - 2] == '/' || name[n - 2] == '#') && (name[n - 1] == '1' || name[n - 1] == '2'))
    return name.substr(0, n >= 3 ? n - 3 : 0);
  return name;
}

}  // namespace

extern "C" {

const char *miso_aln_last_error(void) { return g_err.c_str(); }

int miso_aln_open(const char *path, int n_threads, miso_alnfile_t **out) {
  if (!path || !out) return fail(MISO_EINVAL, "miso_aln_open: null argument");
  *out = nullptr;
  Mapped m;
  m.fd = open(path, O_RDONLY);
  if (m.fd < 0) return fail(MISO_FAILURE, std::string("cannot open ") + path + ": " + std::strerror(errno));
  struct stat st;
  if (fstat(m.fd, &st) != 0) return fail(MISO_FAILURE, std::string("cannot stat ") + path);
  m.len = static_cast<size_t>(st.st_size);
  if (m.len) {
    void *p = mmap(nullptr, m.len, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (p == MAP_FAILED) { m.len = 0; return fail(MISO_FAILURE, std::string("cannot map ") + path); }
    m.p = static_cast<const unsigned char *>(p);
  }
  miso_alnfile *f = nullptr;
  try {
    f = new miso_alnfile();
    int rc;
    if (m.len >= 2 && m.p[0] == 31 && m.p[1] == 139) {
      f->is_bam = true;
      std::vector<unsigned char> data;
      const int T = n_threads > 0 ? n_threads : usable_threads();
      const bool timing = std::getenv("MISO_TIMING") != nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      rc = inflate_all(m, T, data);
      const auto t1 = std::chrono::steady_clock::now();
      if (rc == 0) rc = parse_bam(data, *f, T);
      const auto t2 = std::chrono::steady_clock::now();
      if (timing)
        std::fprintf(stderr, "miso_aln_open: %d threads, inflate %.3f s (%.0f MB), records %.3f s\n", T,
                     std::chrono::duration<double>(t1 - t0).count(), data.size() / 1e6,
                     std::chrono::duration<double>(t2 - t1).count());
    } else {
      rc = parse_sam_parallel(m.p, m.len, *f, n_threads > 0 ? n_threads : usable_threads());
    }
    if (rc != 0) { delete f; return rc; }
    const auto t3 = std::chrono::steady_clock::now();
    f->build_index();
    if (std::getenv("MISO_TIMING"))
      std::fprintf(stderr, "miso_aln_open: index %.3f s\n",
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t3).count());
  } catch (const std::bad_alloc &) {
    delete f;
    return fail(MISO_ENOMEM, "out of memory reading the alignment file");
  }
  *out = f;
  return 0;
}

void miso_aln_close(miso_alnfile_t *f) { delete f; }

int miso_aln_columns(const miso_alnfile_t *f, miso_aln_columns_t *c) {
  if (!f || !c) return fail(MISO_EINVAL, "miso_aln_columns: null argument");
  c->n = f->n();
  c->ref_id = f->ref_id.data(); c->pos = f->pos.data(); c->end = f->end.data();
  c->flag = f->flag.data(); c->l_seq = f->l_seq.data();
  c->cigar_off = f->cigar_off.data(); c->cigar = f->cigar.data();
  c->name_off = f->name_off.data(); c->names = f->names.data();
  return 0;
}

int miso_aln_n_refs(const miso_alnfile_t *f) { return f ? static_cast<int>(f->ref_names.size()) : 0; }

const char *miso_aln_ref_name(const miso_alnfile_t *f, int ref) {
  if (!f || ref < 0 || ref >= static_cast<int>(f->ref_names.size())) return nullptr;
  return f->ref_names[ref].c_str();
}

int64_t miso_aln_ref_length(const miso_alnfile_t *f, int ref) {
  if (!f || ref < 0 || ref >= static_cast<int>(f->ref_len.size())) return -1;
  return f->ref_len[ref];
}

int miso_aln_ref_id(const miso_alnfile_t *f, const char *name) {
  if (!f || !name) return -1;
  auto it = f->ref_index.find(name);
  return it == f->ref_index.end() ? -1 : it->second;
}

int miso_aln_is_bam(const miso_alnfile_t *f) { return f && f->is_bam ? 1 : 0; }

int miso_aln_fetch(const miso_alnfile_t *f, int ref, int64_t start, int64_t end, int64_t *idx,
                   int64_t cap, int64_t *n) {
  if (!f || !n) return fail(MISO_EINVAL, "miso_aln_fetch: null argument");
  int64_t k = 0;
  f->for_overlaps(ref, start, end, [&](int64_t i) {
    if (idx && k < cap) idx[k] = i;
    k++;
  });
  *n = k;
  return 0;
}

}  // extern "C"

// one pass, growing buffers: what miso_batch_add_event_aln (capi.hip) uses directly
int miso_aln_collect_reads(const miso_alnfile_t *f, int ref, int64_t start, int64_t end, int paired,
                           int strand_rule, int target_strand, int given_read_len,
                           std::vector<int32_t> &pos_out, std::string &cig_out, int64_t *n_reads,
                           int64_t *n_strand_discarded) {
  if (!f || !n_reads) return fail(MISO_EINVAL, "miso_aln_parse_reads: null argument");
  if (strand_rule != MISO_STRAND_UNSTRANDED && strand_rule != MISO_STRAND_FIRSTSTRAND)
    return fail(MISO_EINVAL, "miso_aln_parse_reads: unknown strand rule");
  pos_out.clear(); cig_out.clear();
  int64_t kept = 0, discarded = 0;
  const bool check_strand = strand_rule == MISO_STRAND_FIRSTSTRAND && target_strand != 0;  // sam_utils.py:385-390
  auto minus = [&](int64_t i) { return (f->flag[i] & 16) != 0; };
  auto has_cigar = [&](int64_t i) { return f->cigar_off[i + 1] > f->cigar_off[i]; };
  try {
    if (!paired) {
      f->for_overlaps(ref, start, end, [&](int64_t i) {
        if (!has_cigar(i)) return;                                           // sam_utils.py:419
        if (given_read_len > 0 && f->l_seq[i] != given_read_len) return;     // :423-426
        if (check_strand) {                                                  // :353-358
          const char s = minus(i) ? '-' : '+';
          if (s != static_cast<char>(target_strand)) { discarded++; return; }
        }
        pos_out.push_back(f->pos[i]);
        append_cigar_string(*f, i, cig_out);
        kept++;
      });
    } else {
      // pair_sam_reads (sam_utils.py:207-300), names in order of first appearance
      struct Group { int64_t r[2]; int count; };
      std::vector<Group> groups;
      std::unordered_map<std::string_view, size_t> by_name;
      f->for_overlaps(ref, start, end, [&](int64_t i) {
        const int fl = f->flag[i];
        // QC fail, unmapped, mate unmapped or not paired: never enters the pairing (:225-230)
        if ((fl & 0x200) || (fl & 0x4) || (fl & 0x8) || !(fl & 0x1)) return;
        const std::string_view name = strip_mate_id(
            std::string_view(f->names.data() + f->name_off[i], f->name_off[i + 1] - f->name_off[i]));
        auto it = by_name.find(name);
        if (it == by_name.end()) {
          by_name.emplace(name, groups.size());
          groups.push_back({{i, -1}, 1});
          return;
        }
        Group &g = groups[it->second];
        if (g.count < 2) g.r[g.count] = i;
        g.count++;
        if (g.count == 2 && strand_rule == MISO_STRAND_FIRSTSTRAND) {        // :236-248
          if ((f->flag[g.r[0]] & 0x40) && minus(g.r[0])) std::swap(g.r[0], g.r[1]);
          if ((f->flag[g.r[0]] & 0x80) && minus(g.r[0])) std::swap(g.r[0], g.r[1]);
        }
      });
      for (const Group &g : groups) {
        if (g.count != 2) continue;                                          // :255-261
        const int64_t a = g.r[0], b = g.r[1];
        if (minus(a) == minus(b)) continue;                                  // :264-271 same strand
        if (check_strand) {                                                  // :337-346
          bool ok = false;
          if (target_strand == '+') ok = !minus(a);
          else if (target_strand == '-') ok = minus(b);
          if (!ok) { discarded++; continue; }
        }
        if (!has_cigar(a) || !has_cigar(b)) continue;                        // :394-395
        if (given_read_len > 0 && (f->l_seq[a] != given_read_len || f->l_seq[b] != given_read_len))
          continue;                                                          // :399-404
        pos_out.push_back(f->pos[a]);
        pos_out.push_back(f->pos[b]);
        append_cigar_string(*f, a, cig_out);
        append_cigar_string(*f, b, cig_out);
        kept++;
      }
    }
  } catch (const std::bad_alloc &) {
    return fail(MISO_ENOMEM, "out of memory collecting the reads of an event");
  }
  *n_reads = kept;
  if (n_strand_discarded) *n_strand_discarded = discarded;
  return 0;
}


extern "C" int miso_aln_parse_reads(const miso_alnfile_t *f, int ref, int64_t start, int64_t end, int paired,
                                    int strand_rule, int target_strand, int given_read_len,
                                    int32_t *positions, int64_t pos_cap, char *cigar_buf, int64_t cigar_cap,
                                    int64_t *n_reads, int64_t *cigar_bytes, int64_t *n_strand_discarded) {
  if (!f || !n_reads || !cigar_bytes) return fail(MISO_EINVAL, "miso_aln_parse_reads: null argument");
  thread_local std::vector<int32_t> pos_out;
  thread_local std::string cig_out;
  const int rc = miso_aln_collect_reads(f, ref, start, end, paired, strand_rule, target_strand,
                                        given_read_len, pos_out, cig_out, n_reads, n_strand_discarded);
  if (rc) return rc;
  *cigar_bytes = static_cast<int64_t>(cig_out.size());
  if (positions && !pos_out.empty()) std::memcpy(positions, pos_out.data(), 4 * static_cast<size_t>(std::min<int64_t>(pos_cap, pos_out.size())));
  if (cigar_buf && !cig_out.empty()) std::memcpy(cigar_buf, cig_out.data(), static_cast<size_t>(std::min<int64_t>(cigar_cap, cig_out.size())));
  return 0;
}

