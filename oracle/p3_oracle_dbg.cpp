// oracle/p3_oracle_dbg.cpp -- TEST INFRASTRUCTURE ONLY (same rules as p3_oracle.cpp).
//
// CPU restatement of the de Bruijn stage: node set, junction / joint /
// straight classification, node coverage and the GFA text of reference src/DeBruijnGraph.cpp:93-544.
// It restates WHAT the reference's seeded, order-dependent traversal produces as an order-independent
// structure (validated against the reference's own GFA in canonical form, tests/test_oracle_vs_ref.py):
//
//   nodes      = closure of the seeds under Bloom-positive adjacency (MakeDBG/SearchNode/Extend*, :93-297:
//                every reached k-mer's 4 left and 4 right neighbours are Bloom-queried in canonical form, :317-345)
//   junction   = node whose (left degree, right degree) != (1,1), or a (1,1) node alone between two such nodes
//                (:167-179, :212-216)
//   straight   = maximal run of >= 2 consecutive (1,1) nodes; its two end k-mers are the joints (:217-223, :373-391)
//   coverage   = CountNodeCoverage (:393-449) with its quirks (both strands counted separately; the base after the
//                first window read even when the read has exactly k bases)
//   GFA        = PrintGraph (:451-544): S lines, then for every junction the L lines of its read-supported,
//                Bloom-positive neighbours
// Node ids and line order are ours; parity is defined on the ID-free canonical form (tests/gfa_canon.py).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <deque>
#include <string>
#include <unordered_map>
#include <vector>

#include "p3_oracle.h"

typedef unsigned __int128 u128;

namespace {

// Keys beyond two words: N 64-bit words, word 0 least significant, with the handful of operators the restatement below
// uses.  The whole restatement is a template over the key type: u128 for k <= 64, UW<4> up to k = 128, and UW<16>, <32>,
// <63>, <94> for the widths of the reference's long-read cases k = 501, 1001, 2001, 3001 (src/Assemble.cpp:42-49) -- a
// key of fewer words lives in the next wider type (the mask of 2k bits keeps its upper words zero).
template <int N>
struct UW {
  uint64_t w[N];
  UW() { std::memset(w, 0, sizeof w); }
  UW(uint64_t v) {  // NOLINT (implicit on purpose: small constants)
    std::memset(w, 0, sizeof w);
    w[0] = v;
  }
  friend UW operator<<(const UW& a, int n) {
    UW o;
    const int ws = n >> 6, bs = n & 63;
    for (int t = N - 1; t >= 0; t--) {
      uint64_t v = 0;
      if (t - ws >= 0) v = a.w[t - ws] << bs;
      if (bs && t - ws - 1 >= 0) v |= a.w[t - ws - 1] >> (64 - bs);
      o.w[t] = v;
    }
    return o;
  }
  friend UW operator>>(const UW& a, int n) {
    UW o;
    const int ws = n >> 6, bs = n & 63;
    for (int t = 0; t < N; t++) {
      uint64_t v = 0;
      if (t + ws < N) v = a.w[t + ws] >> bs;
      if (bs && t + ws + 1 < N) v |= a.w[t + ws + 1] << (64 - bs);
      o.w[t] = v;
    }
    return o;
  }
  friend UW operator|(const UW& a, const UW& b) {
    UW o;
    for (int t = 0; t < N; t++) o.w[t] = a.w[t] | b.w[t];
    return o;
  }
  friend UW operator&(const UW& a, const UW& b) {
    UW o;
    for (int t = 0; t < N; t++) o.w[t] = a.w[t] & b.w[t];
    return o;
  }
  friend bool operator==(const UW& a, const UW& b) { return std::memcmp(a.w, b.w, sizeof a.w) == 0; }
  friend bool operator<(const UW& a, const UW& b) {
    for (int t = N - 1; t >= 0; t--)
      if (a.w[t] != b.w[t]) return a.w[t] < b.w[t];
    return false;
  }
  friend bool operator<=(const UW& a, const UW& b) { return !(b < a); }
};
typedef UW<4> U256;
constexpr int MAXW = 96;  // words of the widest key + slack (k = 3001 takes 94)

inline uint64_t low64(const u128& v) { return (uint64_t)v; }
template <int N>
inline uint64_t low64(const UW<N>& v) { return v.w[0]; }
inline void to_words(const u128& v, uint64_t* w) {  // (w: MAXW words)
  std::memset(w, 0, 8 * MAXW);
  w[0] = (uint64_t)v;
  w[1] = (uint64_t)(v >> 64);
}
template <int N>
inline void to_words(const UW<N>& v, uint64_t* w) {
  std::memset(w, 0, 8 * MAXW);
  std::memcpy(w, v.w, sizeof v.w);
}
inline void from_words(const uint64_t* w, int nw, u128& v) {
  v = w[0];
  if (nw == 2) v |= (u128)w[1] << 64;
}
template <int N>
inline void from_words(const uint64_t* w, int nw, UW<N>& v) {
  v = UW<N>();
  for (int t = 0; t < nw; t++) v.w[t] = w[t];
}
template <typename K>
struct MaskOf {
  static K get(int k) {
    K m;
    for (int b = 0; b < 2 * k; b++) m.w[b >> 6] |= 1ULL << (b & 63);
    return m;
  }
};
template <>
struct MaskOf<u128> {
  static u128 get(int k) { return (2 * k == 128) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }
};
template <typename K>
inline K mask_of(int k) { return MaskOf<K>::get(k); }

struct HKey {
  size_t operator()(const u128& v) const {
    uint64_t a = (uint64_t)v, b = (uint64_t)(v >> 64);
    a ^= b * 0x9E3779B97F4A7C15ULL;
    a ^= a >> 29;
    a *= 0xBF58476D1CE4E5B9ULL;
    a ^= a >> 32;
    return (size_t)a;
  }
  template <int N>
  size_t operator()(const UW<N>& v) const {
    uint64_t a = v.w[0];
    for (int t = 1; t < N; t++) {
      a ^= v.w[t] * 0x9E3779B97F4A7C15ULL;
      a ^= a >> 29;
      a *= 0xBF58476D1CE4E5B9ULL;
    }
    a ^= a >> 32;
    return (size_t)a;
  }
};

template <typename K>
struct Ctx {
  int k, H, nw;
  uint64_t m;
  const uint64_t* bloom;
  K mask;
  K rc(K x) const {
    K o = 0;
    for (int i = 0; i < k; i++) {
      o = (o << 2) | K((uint64_t)(3 - (unsigned)(low64(x >> (2 * i)) & 3)));
    }
    return o;
  }
  K canon(K x) const {
    K r = rc(x);
    return x <= r ? x : r;
  }
  bool recorded(K x) const {  // IsRecorded, src/DeBruijnGraph.cpp:317-323
    K c = canon(x);
    uint64_t w[MAXW];
    to_words(c, w);
    uint8_t out;
    p3o_bloom_query(bloom, m, H, w, 1, k, &out);
    return out != 0;
  }
  K left_of(K x, unsigned b) const { return (x >> 2) | (K((uint64_t)b) << (2 * k - 2)); }
  K right_of(K x, unsigned b) const { return ((x << 2) | K((uint64_t)b)) & mask; }
  std::string str(K x) const {
    static const char b2c[4] = {'A', 'C', 'G', 'T'};
    std::string s((size_t)k, 'A');
    for (int i = 0; i < k; i++) s[(size_t)i] = b2c[(unsigned)(low64(x >> (2 * (k - 1 - i))) & 3)];
    return s;
  }
};

template <typename K>
struct Node {
  uint8_t nl = 0, nr = 0;
  K l[4], r[4];          // neighbours in the node's canonical orientation
  int junction_id = 0;   // > 0: junction
  int straight_id = 0;   // > 0: member of that straight
  bool joint = false;
  // junction counters (canonical orientation)
  int coverage = 0, lcov[4] = {0, 0, 0, 0}, rcov[4] = {0, 0, 0, 0};
};

inline unsigned fcode(char c) { return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0; }
inline unsigned ccode(char c) { return c == 'A' ? 3 : c == 'C' ? 2 : c == 'G' ? 1 : 0; }  // comp. code; unknown -> 0

}  // namespace

extern "C" {

void p3o_free(char* p) { delete[] p; }

}  // extern "C"

// Returns the GFA text (reference format) in a new[]-allocated buffer (*out_text, *out_len); 0 on success.
// joints_out (optional): "KMER\tcoverage\n" per joint, k-mer in its smaller orientation, sorted -- JointInfo::coverage
// (src/DeBruijnGraph.cpp:24-28,441-449), which the reference fills but never prints.
template <typename K>
static int dbg_impl_t(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len, int k,
                    const uint64_t* bloom_words, uint64_t m, int H, const uint64_t* seed_keys, uint64_t n_seeds,
                    char** out_text, uint64_t* out_len, uint64_t* n_junctions, uint64_t* n_joints, uint64_t* n_straights,
                    std::string* joints_out) {
  typedef Node<K> Node;
  typedef std::unordered_map<K, Node, HKey> NodeMap;
  Ctx<K> cx;
  cx.k = k;
  cx.H = H;
  cx.m = m;
  cx.bloom = bloom_words;
  cx.nw = (2 * k + 63) / 64;
  cx.mask = mask_of<K>(k);
  NodeMap nodes;
  // ---- closure of the seeds (G4) ----
  std::deque<K> q;
  for (uint64_t i = 0; i < n_seeds; i++) {
    K s;
    from_words(seed_keys + i * cx.nw, cx.nw, s);
    q.push_back(s);
  }
  while (!q.empty()) {
    K c = cx.canon(q.front());
    q.pop_front();
    if (nodes.count(c)) continue;
    Node n;
    for (unsigned b = 0; b < 4; b++) {
      K a = cx.left_of(c, b);
      if (cx.recorded(a)) n.l[n.nl++] = a;
    }
    for (unsigned b = 0; b < 4; b++) {
      K a = cx.right_of(c, b);
      if (cx.recorded(a)) n.r[n.nr++] = a;
    }
    nodes.emplace(c, n);
    for (int i = 0; i < n.nl; i++) q.push_back(n.l[i]);
    for (int i = 0; i < n.nr; i++) q.push_back(n.r[i]);
  }
  auto is11 = [&](K oriented) {
    const Node& n = nodes.at(cx.canon(oriented));
    return n.nl == 1 && n.nr == 1;
  };
  // oriented right / left neighbour of an oriented (1,1) node
  auto right1 = [&](K x) {
    K c = cx.canon(x);
    const Node& n = nodes.at(c);
    return x == c ? n.r[0] : cx.rc(n.l[0]);
  };
  auto left1 = [&](K x) {
    K c = cx.canon(x);
    const Node& n = nodes.at(c);
    return x == c ? n.l[0] : cx.rc(n.r[0]);
  };
  // ---- classification: junctions, straights, joints ----
  int junction_id = 0, straight_id = 0;
  struct Straight {
    std::string seq;
    K left_joint, right_joint;  // oriented as in seq
  };
  std::vector<Straight> straights;
  std::unordered_map<K, int, HKey> joint_of;  // oriented joint k-mer (as stored) -> straight id
  std::vector<K> order;
  order.reserve(nodes.size());
  for (auto& kv : nodes) order.push_back(kv.first);
  std::sort(order.begin(), order.end());  // deterministic ids
  for (K c : order) {
    Node& n = nodes[c];
    if (!(n.nl == 1 && n.nr == 1)) n.junction_id = ++junction_id;
  }
  const size_t guard = nodes.size() + 2;
  for (K c : order) {
    Node& n = nodes[c];
    if (n.junction_id || n.straight_id) continue;
    // walk left to the start of the run (orientation: canonical orientation of c)
    K start = c;
    size_t steps = 0;
    while (steps++ < guard) {
      K p = left1(start);
      if (!is11(p) || cx.canon(p) == cx.canon(c)) break;
      start = p;
    }
    // walk right collecting the run
    std::vector<K> run;
    run.push_back(start);
    steps = 0;
    while (steps++ < guard) {
      K nx = right1(run.back());
      if (!is11(nx) || cx.canon(nx) == cx.canon(start)) break;
      run.push_back(nx);
    }
    if (run.size() == 1) {
      n.junction_id = ++junction_id;  // "this node cannot extend" (:212-216)
      continue;
    }
    Straight st;
    st.seq = cx.str(run[0]);
    static const char b2c[4] = {'A', 'C', 'G', 'T'};
    for (size_t i = 1; i < run.size(); i++) st.seq += b2c[(unsigned)(low64(run[i]) & 3)];
    st.left_joint = run.front();
    st.right_joint = run.back();
    straights.push_back(st);
    ++straight_id;
    for (K x : run) nodes[cx.canon(x)].straight_id = straight_id;
    nodes[cx.canon(run.front())].joint = true;
    nodes[cx.canon(run.back())].joint = true;
    joint_of[run.front()] = straight_id;
    joint_of[run.back()] = straight_id;
  }
  // ---- coverage (G5): junction keys are stored in canonical orientation here ----
  auto junction_at = [&](K key) -> Node* {  // exact-orientation lookup, like junctions.find(key)
    auto it = nodes.find(key);
    if (it == nodes.end() || !it->second.junction_id) return nullptr;
    return &it->second;  // map key == canonical == stored orientation
  };
  for (uint64_t r = 0; r < n_reads; r++) {
    const char* s = bases + offsets[r];
    const uint64_t L = offsets[r + 1] - offsets[r];
    if (L < min_len || L < (uint64_t)k) continue;
    K fw = 0, bw = 0;
    for (uint64_t i = 0; i < L; i++) {
      fw = ((fw << 2) | K((uint64_t)fcode(s[i]))) & cx.mask;
      bw = (bw >> 2) | (K((uint64_t)ccode(s[i])) << (2 * k - 2));
      if (i + 1 < (uint64_t)k) continue;
      Node* jf = junction_at(fw);
      Node* jb = junction_at(bw);
      if (jf) jf->coverage++;
      if (jb) jb->coverage++;
      {  // joints.find(kmer_Fw) / joints.find(kmer_Bw) (:445-447): the map holds the joint in ONE orientation, so a
         // window counts once -- twice when it is its own reverse complement
        auto it = nodes.find(fw < bw ? fw : bw);
        if (it != nodes.end() && it->second.joint) it->second.coverage += (fw == bw) ? 2 : 1;
      }
      const char nxt = (i + 1 < L) ? s[i + 1] : '\0';
      if (i + 1 == (uint64_t)k) {  // first window (:407-413): reads read[k] unguarded
        if (jf) jf->rcov[fcode(nxt)]++;
        else if (jb) jb->lcov[ccode(nxt)]++;
      } else {
        const char prv = s[i - k];
        if (jf) {
          jf->lcov[fcode(prv)]++;
          if (i < L - 1) jf->rcov[fcode(nxt)]++;
        } else if (jb) {
          jb->rcov[ccode(prv)]++;
          if (i < L - 1) jb->lcov[ccode(nxt)]++;
        }
      }
    }
  }
  // ---- GFA text (G6) ----
  std::string out = "H\tVN:Z:1.0\n";
  for (size_t i = 0; i < straights.size(); i++)
    out += "S\tStraight_" + std::to_string(i + 1) + "\t" + straights[i].seq + "\tKC:i:" + std::to_string(straights[i].seq.size()) + "\n";
  for (K c : order) {
    const Node& n = nodes[c];
    if (n.junction_id)
      out += "S\tJunction_" + std::to_string(n.junction_id) + "\t" + cx.str(c) + "\tKC:i:" + std::to_string((long long)n.coverage * k) + "\n";
  }
  const std::string ov = std::to_string(k - 1) + "M\n";
  auto name_of = [&](K x, std::string& nm, char& orient) -> bool {
    // junctions.find(x) / joints.find(x), then the complement (:474-505)
    if (Node* j = junction_at(x)) {
      nm = "Junction_" + std::to_string(j->junction_id);
      orient = '+';
      return true;
    }
    auto it = joint_of.find(x);
    if (it != joint_of.end()) {
      nm = "Straight_" + std::to_string(it->second);
      orient = '+';
      return true;
    }
    K y = cx.rc(x);
    if (Node* j = junction_at(y)) {
      nm = "Junction_" + std::to_string(j->junction_id);
      orient = '-';
      return true;
    }
    it = joint_of.find(y);
    if (it != joint_of.end()) {
      nm = "Straight_" + std::to_string(it->second);
      orient = '-';
      return true;
    }
    return false;
  };
  for (K c : order) {
    const Node& n = nodes[c];
    if (!n.junction_id) continue;
    const std::string me = "Junction_" + std::to_string(n.junction_id);
    for (unsigned b = 0; b < 4; b++) {
      if (n.lcov[b] == 0) continue;
      K a = cx.left_of(c, b);
      if (!cx.recorded(a)) continue;
      std::string nm;
      char o;
      if (name_of(a, nm, o)) out += "L\t" + nm + "\t" + o + "\t" + me + "\t+\t" + ov;
    }
    for (unsigned b = 0; b < 4; b++) {
      if (n.rcov[b] == 0) continue;
      K a = cx.right_of(c, b);
      if (!cx.recorded(a)) continue;
      std::string nm;
      char o;
      if (name_of(a, nm, o)) out += "L\t" + me + "\t+\t" + nm + "\t" + o + "\t" + ov;
    }
  }
  char* buf = new char[out.size() + 1];
  std::memcpy(buf, out.data(), out.size() + 1);
  *out_text = buf;
  *out_len = out.size();
  if (n_junctions) *n_junctions = (uint64_t)junction_id;
  if (n_joints) *n_joints = joint_of.size();
  if (n_straights) *n_straights = straights.size();
  if (joints_out) {
    std::vector<std::pair<std::string, int>> jl;
    for (auto& kv : nodes)
      if (kv.second.joint) jl.push_back({cx.str(kv.first), kv.second.coverage});  // (map key = canonical = smaller)
    std::sort(jl.begin(), jl.end());
    for (auto& j : jl) *joints_out += j.first + "\t" + std::to_string(j.second) + "\n";
  }
  return 0;
}

static int dbg_impl(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len, int k,
                    const uint64_t* bloom_words, uint64_t m, int H, const uint64_t* seed_keys, uint64_t n_seeds,
                    char** out_text, uint64_t* out_len, uint64_t* n_junctions, uint64_t* n_joints, uint64_t* n_straights,
                    std::string* joints_out) {
  if (k < 21 || k > 3001) return 2;
#define P3O_DBG_AS(K)                                                                                                      \
  return dbg_impl_t<K>(bases, offsets, n_reads, min_len, k, bloom_words, m, H, seed_keys, n_seeds, out_text, out_len, \
                       n_junctions, n_joints, n_straights, joints_out)
  const int nw = (2 * k + 63) / 64;
  if (nw <= 2) P3O_DBG_AS(u128);
  if (nw <= 4) P3O_DBG_AS(U256);
  if (nw <= 16) P3O_DBG_AS(UW<16>);
  if (nw <= 32) P3O_DBG_AS(UW<32>);
  if (nw <= 63) P3O_DBG_AS(UW<63>);
  P3O_DBG_AS(UW<94>);
#undef P3O_DBG_AS
}

extern "C" {

int p3o_dbg(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len, int k,
            const uint64_t* bloom_words, uint64_t m, int H, const uint64_t* seed_keys, uint64_t n_seeds,
            char** out_text, uint64_t* out_len, uint64_t* n_junctions, uint64_t* n_joints, uint64_t* n_straights) {
  return dbg_impl(bases, offsets, n_reads, min_len, k, bloom_words, m, H, seed_keys, n_seeds, out_text, out_len, n_junctions,
                  n_joints, n_straights, nullptr);
}

// as p3o_dbg, plus the joint list as text (*joints_text, new[]-allocated)
int p3o_dbg_joints(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t min_len, int k,
                   const uint64_t* bloom_words, uint64_t m, int H, const uint64_t* seed_keys, uint64_t n_seeds,
                   char** out_text, uint64_t* out_len, char** joints_text, uint64_t* joints_len) {
  std::string jt;
  int rc = dbg_impl(bases, offsets, n_reads, min_len, k, bloom_words, m, H, seed_keys, n_seeds, out_text, out_len, nullptr, nullptr,
                    nullptr, &jt);
  if (rc) return rc;
  char* buf = new char[jt.size() + 1];
  std::memcpy(buf, jt.data(), jt.size() + 1);
  *joints_text = buf;
  *joints_len = jt.size();
  return 0;
}

}  // extern "C"
