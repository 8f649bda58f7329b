// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Literal single-threaded CPU restatement of BreakID's hot path over
// the same columnar record table the product takes (include/breakid_hip.h).  It exists to check the
// HIP implementation and to serve as bench.py's `cpu_baseline` (kind "port").  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product never does.
//
// Parity status: PINNED — every function below is checked against the real reference compiled from
// /root/reference by oracle/Makefile (`make -C oracle ref` -> oracle/_ref/ref_harness, ref_units) and
// against the golden vectors those tools produced (tests/golden/, generator tools/make_golden.py).
//
// Each function cites the reference lines it restates (paths relative to /root/reference/src).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../include/breakid_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// CIGAR model: CigarRoller.cc:26-46 (operator+=: drop count 0, merge adjacent equal ops, M==X===),
// :67-116 (Add(char,int)), :119-136 (Add(const char*)), :194-205 (Set(uint32*,n)),
// Cigar.cc:80-144 (reference base count, begin/end clips), Cigar.h:215-228 (getNumMatches).
// ops: 1 match, 3 insert, 4 del, 5 skip, 6 softClip, 7 hardClip, 8 pad (Cigar.h:65-76)
struct Roller
{
  std::vector<std::pair<int, uint32_t>> ops;  // (operation, count)
  void clear() { ops.clear(); }
  void add_op(int op, uint32_t count)
  {
    if (count == 0) return;
    if (ops.empty() || ops.back().first != op)
      ops.emplace_back(op, count);
    else
      ops.back().second += count;
  }
  void add_char(int c, int count)
  {
    switch (c)
    {
    case 0: case 'M': add_op(1, (uint32_t) count); break;
    case 1: case 'I': add_op(3, (uint32_t) count); break;
    case 2: case 'D': add_op(4, (uint32_t) count); break;
    case 3: case 'N': add_op(5, (uint32_t) count); break;
    case 4: case 'S': add_op(6, (uint32_t) count); break;
    case 5: case 'H': add_op(7, (uint32_t) count); break;
    case 6: case 'P': add_op(8, (uint32_t) count); break;
    case 7: case '=': add_op(1, (uint32_t) count); break;
    case 8: case 'X': add_op(1, (uint32_t) count); break;
    default: break;  // reference prints an error and ignores the op
    }
  }
  void set_text(const char *s, size_t len)
  {
    clear();
    int count = 0;
    size_t i = 0;
    while (i < len && s[i])
    {
      if (s[i] >= '0' && s[i] <= '9')
      {
        long v = 0;  // strtol, base 10 (no sign/space can follow a digit test)
        while (i < len && s[i] >= '0' && s[i] <= '9')
        {
          v = v * 10 + (s[i] - '0');
          if (v > 0x7fffffffffffffL) v = 0x7fffffffffffffL;
          ++i;
        }
        count = (int) v;
      }
      else
      {
        add_char((unsigned char) s[i], count);
        ++i;
      }
    }
  }
  void set_bam(const uint32_t *w, uint32_t n)
  {
    clear();
    for (uint32_t i = 0; i < n; ++i) add_char((int) (w[i] & 0xF), (int) (w[i] >> 4));
  }
  static char op_char(int op)
  {
    switch (op)
    {
    case 1: case 2: return 'M';
    case 3: return 'I';
    case 4: return 'D';
    case 5: return 'N';
    case 6: return 'S';
    case 7: return 'H';
    case 8: return 'P';
    }
    return '?';
  }
  std::string str() const
  {
    std::string s;
    for (auto &o : ops) s += std::to_string(o.second) + op_char(o.first);
    return s;
  }
  int reflen() const
  {
    int n = 0;
    for (auto &o : ops)
      if (o.first == 1 || o.first == 2 || o.first == 4 || o.first == 5) n += (int) o.second;
    return n;
  }
  int begin_clips() const
  {
    int n = 0;
    for (auto &o : ops)
    {
      if (o.first == 6 || o.first == 7) n += (int) o.second; else break;
    }
    return n;
  }
  int end_clips() const
  {
    int n = 0;
    for (size_t i = ops.size(); i-- > 0;)
    {
      if (ops[i].first == 6 || ops[i].first == 7) n += (int) ops[i].second; else break;
    }
    return n;
  }
  int matches() const
  {
    int n = 0;
    for (auto &o : ops) if (o.first == 1) n += (int) o.second;
    return n;
  }
};

// full match of ([0-9]+[MS]){2}  (CigarRoller.cc:326)
bool two_op_ms(const char *s, size_t len)
{
  size_t i = 0;
  for (int k = 0; k < 2; ++k)
  {
    size_t d = i;
    while (i < len && s[i] >= '0' && s[i] <= '9') ++i;
    if (i == d) return false;
    if (i >= len || (s[i] != 'M' && s[i] != 'S')) return false;
    ++i;
  }
  return i == len;
}

// CigarRoller::is_complementary_cigar, CigarRoller.cc:323-346
bool is_complementary(const Roller &c1, const char *c2, size_t c2len, int e)
{
  std::string s1 = c1.str();
  Roller r2;
  r2.set_text(c2, c2len);
  if (!two_op_ms(s1.data(), s1.size()) || !two_op_ms(c2, c2len)) return false;
  int c1_m = c1.matches(), c2_m = r2.matches();
  int c1_s = c1.begin_clips() + c1.end_clips();
  int c2_s = r2.end_clips() + r2.begin_clips();
  return (c1_m <= c2_s + e && c1_m >= c2_s - e) && (c1_m + c1_s == c2_m + c2_s);
}

// 64-bit code of a CIGAR text as the bk_split fields carry it (include/breakid_hip.h): texts of the form <n><M|S><n><M|S>
// are encoded exactly (bit 63 set), anything else as a 63-bit FNV-1a of the text.  The reference compares the strings.
uint64_t cigar_code(const char *t, size_t len)
{
  const unsigned char *s = (const unsigned char *) t;
  size_t i = 0;
  uint64_t code = 1ull << 63;
  bool exact = true;
  for (int k = 0; k < 2 && exact; ++k)
  {
    uint32_t z = 0, d = 0;
    while (i < len && s[i] == '0') { ++z; ++i; }
    uint64_t v = 0;
    while (i < len && s[i] >= '0' && s[i] <= '9' && d < 10) { v = v * 10 + (uint64_t) (s[i] - '0'); ++i; ++d; }
    if ((z == 0 && d == 0) || z > 3 || v >= (1ull << 28) || i >= len || (s[i] != 'M' && s[i] != 'S')) { exact = false; break; }
    const uint64_t op = s[i] == 'S' ? 1ull : 0ull;
    ++i;
    code |= k == 0 ? ((uint64_t) z << 60) | (op << 57) | (v << 28) : ((uint64_t) z << 58) | (op << 56) | v;
  }
  if (exact && i == len) return code;
  uint64_t h = 0xCBF29CE484222325ull;
  for (size_t j = 0; j < len; ++j) { h ^= s[j]; h *= 0x100000001B3ull; }
  return h & ~(1ull << 63);
}

// the ABI's second 32-bit name hash (include/breakid_hip.h: bk_qname_check), restated: bk_split.reserved carries it for SA contig
// names that are not known names
uint32_t name_check32(const char *s, size_t len)
{
  uint32_t h = 0x811C9DC5u ^ ((uint32_t) len * 0x9E3779B1u);
  for (size_t i = 0; i < len; ++i)
  {
    h = (h ^ (uint8_t) s[i]) * 0x01000193u;
    h = (h << 13) | (h >> 19);
    h = h * 5u + 0xE6546B64u;
  }
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h ? h : 1u;
}
uint64_t text_hash(const char *s, size_t len)  // FNV-1a 64: ids of contig names that are not in the header
{
  uint64_t h = 0xCBF29CE484222325ull;
  for (size_t i = 0; i < len; ++i)
  {
    h ^= (unsigned char) s[i];
    h *= 0x100000001B3ull;
  }
  return h;
}

// util_bam.cc:128-142
std::string chrom_id_to_name(int tid)
{
  if (tid == 23) return "chrY";
  if (tid == 22) return "chrX";
  if (tid >= 0 && tid < 22) return "chr" + std::to_string(tid + 1);
  return "";
}

struct XY
{
  uint32_t x, y, id;
  int32_t cluster;
  int k1, k2;  // fast path cluster_id "k1:k2"
};

// BreakID.cc:1813-1877 mask_pairs_chr_pos (quirks H4)
template <class T> void mask_pairs(std::vector<T> &v, long distance)
{
  long np = (long) v.size();
  std::vector<T> out;
  if (np <= 2)
  {
    v.clear();
    return;
  }
  long Lx = std::labs((long) (int32_t) (v[1].x - v[2].x));
  long Ly = std::labs((long) (int32_t) (v[1].y - v[2].y));
  if (!(Lx > distance || Ly > distance)) out.push_back(v[1]);
  for (long i = 1; i < np - 1; ++i)
  {
    long ll = std::labs((long) (int32_t) (v[i - 1].x - v[i].x));
    long lr = std::labs((long) (int32_t) (v[i + 1].x - v[i].x));
    Lx = ll < lr ? ll : lr;
    ll = std::labs((long) (int32_t) (v[i - 1].y - v[i].y));
    lr = std::labs((long) (int32_t) (v[i + 1].y - v[i].y));
    Ly = ll < lr ? ll : lr;
    if (!(Lx > distance || Ly > distance)) out.push_back(v[i]);
  }
  v = out;
}

template <class T> bool cmp_x(T a, T b) { return a.x < b.x; }  // BreakID.h:170-173
template <class T> bool cmp_y(T a, T b) { return a.y < b.y; }  // BreakID.h:175-178

// Diagnostic only (ora_sort_stats): which segments would libstdc++'s introsort hand to its heapsort branch
// (depth limit 2*floor(log2 n), bits/stl_algo.h __introsort_loop)?  Replayed on a COPY with the library's own
// partition step; the result the oracle uses always comes from the real std::sort below.
struct SortStats
{
  uint64_t sorts = 0, heap_segments = 0, heap_elems = 0, max_heap = 0;
};
SortStats g_sort_stats;
bool g_sort_probe = false;
template <class It, class Cmp> void probe_loop(It first, It last, long depth, Cmp c)
{
  while (last - first > 16)
  {
    if (depth == 0)
    {
      uint64_t m = (uint64_t) (last - first);
      g_sort_stats.heap_segments++;
      g_sort_stats.heap_elems += m;
      if (m > g_sort_stats.max_heap) g_sort_stats.max_heap = m;
      return;
    }
    --depth;
    It cut = std::__unguarded_partition_pivot(first, last, __gnu_cxx::__ops::__iter_comp_iter(c));
    probe_loop(cut, last, depth, c);
    last = cut;
  }
}
template <class T, class Cmp> void ref_sort(std::vector<T> &v, Cmp c)
{
  if (g_sort_probe && v.size() > 16)
  {
    std::vector<T> copy(v);
    g_sort_stats.sorts++;
    probe_loop(copy.begin(), copy.end(), 2 * (long) std::__lg((long) copy.size()), c);
  }
  std::sort(v.begin(), v.end(), c);
}

// BreakID.cc:1271-1285 remove_isolated_pairs (distance is truncated to long at :1275)
template <class T> void remove_isolated(std::vector<T> &v, double w)
{
  ref_sort(v, cmp_x<T>);
  mask_pairs(v, (long) w);
  if (!v.empty())
  {
    ref_sort(v, cmp_y<T>);
    mask_pairs(v, (long) w);
    if (!v.empty()) ref_sort(v, cmp_x<T>);
  }
}

// BreakID.cc:1046-1160 find_cluster_pairs_enspan_fast (min_reads = 2)
template <class T> int fast_cluster(std::vector<T> &v, double w, int min_reads)
{
  if (v.empty()) return 0;  // reference reads enspan[0] of an empty vector (UB); callers pass >= 2
  std::vector<int> cl;
  std::vector<T> tmp;
  int k = 1, n = (int) v.size();
  long pre = v[0].x;
  cl.push_back(0);
  for (int i = 1; i < n; ++i)
  {
    if (v[i].x <= pre + w && i != n - 1)
      cl.push_back(i);
    else
    {
      if ((int) cl.size() >= min_reads)
      {
        for (int j : cl)
        {
          v[j].k1 = k;
          tmp.push_back(v[j]);
        }
        ++k;
      }
      pre = v[i].x;
      cl.clear();
      cl.push_back(i);
    }
  }
  v = tmp;
  tmp.clear();
  cl.clear();
  ref_sort(v, cmp_y<T>);
  k = 1;
  n = (int) v.size();
  if (n == 0) return 0;  // reference: UB read of enspan[0]; nothing survives either way
  pre = v[0].y;
  cl.push_back(0);
  for (int i = 1; i < n; ++i)
  {
    if (v[i].y <= pre + w && i != n - 1)
      cl.push_back(i);
    else
    {
      if ((int) cl.size() >= min_reads)
      {
        for (int j : cl)
        {
          v[j].k2 = k;
          tmp.push_back(v[j]);
        }
        ++k;
      }
      pre = v[i].y;
      cl.clear();
      cl.push_back(i);
    }
  }
  v = tmp;
  tmp.clear();
  ref_sort(v, cmp_x<T>);
  std::map<std::pair<int, int>, int> key, key_cl;  // string ids "k1:k2" compare equal iff (k1,k2) equal
  for (auto &p : v) key[{p.k1, p.k2}]++;
  k = 0;
  for (auto &p : v)
  {
    auto it = key.find({p.k1, p.k2});
    if (it->second >= min_reads)
    {
      auto it2 = key_cl.find({p.k1, p.k2});
      if (it2 == key_cl.end())
      {
        ++k;
        p.cluster = k;
        key_cl[{p.k1, p.k2}] = k;
      }
      else
        p.cluster = it2->second;
      tmp.push_back(p);
    }
  }
  v = tmp;
  return k;
}

// ---------------------------------------------------------------------------------------------
// util_cluster.cc: agglomerative clustering, average linkage (literal, N x N matrix).
struct AhcNode
{
  int is_root, num_points, m0, m1;
  std::vector<int> points;
  std::vector<std::pair<int, double>> nb;  // sorted neighbour list (target, distance)
};
struct Ahc
{
  size_t n = 0;
  int num_root = 0;
  std::vector<AhcNode> nodes;
  std::vector<double> mat;
  double d(int a, int b) const { return mat[(size_t) a * n + b]; }
};

// util_cluster.cc:249-297 insert_sorted / insert_before / insert_after
void ahc_insert_sorted(std::vector<std::pair<int, double>> &L, std::pair<int, double> e)
{
  for (size_t i = 0; i + 1 < L.size(); ++i)
  {
    if (L[i].second >= e.second)
    {
      L.insert(L.begin() + i, e);
      return;
    }
  }
  if (L.back().second > e.second)
    L.insert(L.end() - 1, e);
  else
    L.push_back(e);
}

// util_cluster.cc:201-215 average_linkage (sum order a outer, b inner; divide by int m*n)
double ahc_average(const Ahc &c, const std::vector<int> &a, const std::vector<int> &b)
{
  double total = 0.0;
  int m = (int) a.size(), n = (int) b.size();
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) total += c.d(a[i], b[j]);
  return total / (m * n);
}

// util_cluster.cc:112-156 update_neighbours / add_neighbour, :158-199 get_distance
void ahc_update_neighbours(Ahc &c)
{
  int cur = (int) c.nodes.size() - 1;
  int seen = 1, target = cur;
  while (seen < c.num_root)
  {
    --target;
    if (c.nodes[target].is_root)
    {
      ++seen;
      double dist;
      if ((size_t) cur < c.n && (size_t) target < c.n)
        dist = c.d(cur, target);
      else
        dist = ahc_average(c, c.nodes[cur].points, c.nodes[target].points);
      if (!c.nodes[cur].nb.empty())
        ahc_insert_sorted(c.nodes[cur].nb, {target, dist});
      else
        c.nodes[cur].nb.push_back({target, dist});
    }
  }
}

// util_cluster.cc:7-47 init_cluster, :49-84 matrix, :86-110 leaves, :299-396 merge loop
void ahc_run(Ahc &c, const std::vector<uint32_t> &xs, const std::vector<uint32_t> &ys, long threshold)
{
  c.n = xs.size();
  c.mat.assign(c.n * c.n, 0.0);
  for (size_t i = 0; i < c.n; ++i)
    for (size_t j = 0; j < c.n; ++j)
    {
      double dx = (double) xs[i] - (double) xs[j], dy = (double) ys[i] - (double) ys[j];
      c.mat[i * c.n + j] = std::sqrt(std::pow(dx, 2) + std::pow(dy, 2));
    }
  for (size_t i = 0; i < c.n; ++i)
  {
    AhcNode nd;
    nd.is_root = 1;
    nd.num_points = 1;
    nd.m0 = nd.m1 = -1;
    nd.points.push_back((int) c.nodes.size());
    c.nodes.push_back(nd);
    c.num_root++;
    ahc_update_neighbours(c);
  }
  while (c.num_root > 1)
  {
    double best = DBL_MAX;
    int first = -1, second = 0;
    int seen = 0, j = (int) c.nodes.size();
    while (seen < c.num_root)
    {
      --j;
      if (!c.nodes[j].is_root) continue;
      ++seen;
      for (auto &e : c.nodes[j].nb)
      {
        if (c.nodes[e.first].is_root)
        {
          if (first == -1 || e.second < best)
          {
            first = j;
            second = e.first;
            best = e.second;
          }
          break;
        }
      }
    }
    if (first != -1 && best <= threshold)
    {
      AhcNode nd;
      nd.is_root = 1;
      nd.m0 = first;
      nd.m1 = second;
      nd.num_points = c.nodes[first].num_points + c.nodes[second].num_points;
      c.nodes[first].is_root = 0;
      c.nodes[second].is_root = 0;
      nd.points = c.nodes[first].points;
      nd.points.insert(nd.points.end(), c.nodes[second].points.begin(), c.nodes[second].points.end());
      c.nodes.push_back(nd);
      c.num_root--;
      ahc_update_neighbours(c);
    }
    else
      break;
  }
}

// BreakID.cc:1304-1352 find_cluster_pairs_enspan_ahc + add_cluster_id_for_enspan_vec
template <class T> int ahc_cluster(std::vector<T> &v, double w, int min_reads)
{
  std::vector<uint32_t> xs, ys;
  for (auto &p : v)
  {
    xs.push_back(p.x);
    ys.push_back(p.y);
  }
  Ahc c;
  ahc_run(c, xs, ys, (long) w);
  std::vector<T> out;
  int k = 0, roots = 0;
  for (auto &nd : c.nodes)
  {
    if (nd.is_root) ++roots;
    if (nd.is_root && nd.num_points >= min_reads)
    {
      for (int p : nd.points)
      {
        v[p].cluster = k;
        out.push_back(v[p]);
      }
      ++k;
    }
  }
  v = out;
  return roots;
}

// ---------------------------------------------------------------------------------------------
struct Oracle
{
  std::vector<uint32_t> tlen;
  std::vector<std::string> tname;
  std::vector<uint32_t> tprefix;  // util_bam.cc:57-68 running sum (uint32 wrap)
  std::map<std::string, int> name_id;
  bk_soa s{};
  double mean = 0, sd = 0;
  int maxspan = 1;

  std::vector<bk_pair> scan, iso, clustered;
  std::vector<uint64_t> scan_off, iso_off, clustered_off;
  std::vector<int32_t> group_keys;
  std::vector<bk_split> splits;
  std::vector<bk_cluster> clusters;
  std::string err;

  int intern(const std::string &s)
  {
    auto it = name_id.find(s);
    if (it != name_id.end()) return it->second;
    return (int) (0x40000000u | (uint32_t) (text_hash(s.data(), s.size()) & 0x3FFFFFFFu));
  }
  uint32_t gpos(int tid, int32_t pos) const
  {
    uint32_t base = tid <= 0 ? 0u : tprefix[std::min<size_t>((size_t) tid, tprefix.size() - 1)];
    return base + (uint32_t) pos;
  }
  std::string rname(int tid) const { return tid < 0 ? "*" : tname[tid]; }
  int32_t endpos(uint64_t i) const  // sam.c:344-350 bam_endpos
  {
    uint32_t nc = s.cigar_off[i + 1] - s.cigar_off[i];
    if (!(s.flag[i] & 4) && nc > 0)
    {
      int l = 0;
      for (uint32_t k = 0; k < nc; ++k)
      {
        uint32_t w = s.cigar[s.cigar_off[i] + k], op = w & 15;
        if ((0x3C1A7 >> (op << 1)) & 2) l += (int) (w >> 4);
      }
      return s.pos[i] + l;
    }
    return s.pos[i] + 1;
  }
};

// evidence tuple of one record, BreakID.cc:895-1016.  Returns false when the record yields none.
bool record_split(Oracle &o, uint64_t i, bk_split &t)
{
  const bk_soa &s = o.s;
  uint32_t a0 = s.aux_off[i], a1 = s.aux_off[i + 1];
  if (a1 <= a0) return false;  // sa_tag == ""
  uint16_t flag = s.flag[i];
  if ((flag & 0x400) || !(flag & 1)) return false;  // :898
  const char *blob = (const char *) s.aux + a0;
  size_t blen = a1 - a0;
  const char *tab = (const char *) memchr(blob, '\t', blen);
  std::string oc, sa;
  if (tab)
  {
    oc.assign(blob, tab - blob);
    sa.assign(tab + 1, blob + blen - (tab + 1));
  }
  else
    sa.assign(blob, blen);
  if (sa.empty()) return false;
  // util_bed.cc:194-222 split_string(sa, ","), empty tokens dropped
  std::vector<std::string> f;
  {
    size_t st = 0;
    while (true)
    {
      size_t e = sa.find(',', st);
      std::string tok = sa.substr(st, e == std::string::npos ? std::string::npos : e - st);
      if (!tok.empty()) f.push_back(tok);
      if (e == std::string::npos) break;
      st = e + 1;
    }
  }
  if (f.size() < 4) return false;  // reference would index out of range (UB); such SA text is invalid
  Roller sa_c, own, tmp;
  sa_c.set_text(f[3].data(), f[3].size());
  own.set_bam(s.cigar + s.cigar_off[i], s.cigar_off[i + 1] - s.cigar_off[i]);
  if (!oc.empty())
    tmp.set_text(oc.data(), oc.size());
  else
    tmp = own;  // Set(align.getCigarString()) re-parses the rolled text: identical ops
  if (!is_complementary(tmp, f[3].data(), f[3].size(), 10)) return false;
  memset(&t, 0, sizeof t);
  t.rec = i;
  t.tid = s.tid[i];
  t.pos = s.pos[i];
  t.endpos = o.endpos(i);
  t.qhash = s.qhash[i];
  t.qcheck = s.qcheck ? s.qcheck[i] : 0u;
  bool secondary = (flag & 0x100) != 0;
  t.flags = secondary ? 1u : 0u;
  uint32_t sa_start = (uint32_t) atoi(f[1].c_str());  // stoi
  uint32_t sa_end = sa_start + (uint32_t) sa_c.reflen() - 1;  // CigarRoller.cc:316-321
  long a_start = (long) s.pos[i] + 1;  // BamAlignment.cc:172-175
  int own_len = own.reflen();
  long a_end = (own_len == 0 ? (long) s.pos[i] : (long) s.pos[i] + own_len - 1) + 1;  // :107-116,:177-180
  std::string own_str = own.str();
  int own_chr = o.intern(chrom_id_to_name(s.tid[i]));
  int sa_chr = o.intern(f[0]);
  // a contig name that is neither in the header nor chr1..22,X,Y travels as a 30-bit hash id; its second hash rides in `reserved`
  // so that two such names are only equal when 62 bits agree (the reference compares the strings, BreakID.cc:627-637)
  if (sa_chr & 0x40000000) t.reserved = name_check32(f[0].data(), f[0].size());
  uint32_t own_end_val, own_bp = 0, sa_bp = 0;
  uint64_t own_cig;
  bool poison = false;
  if (!oc.empty())
  {
    own_cig = cigar_code(oc.data(), oc.size());
    own_end_val = (uint32_t) ((uint32_t) a_start + (uint32_t) tmp.reflen() - 1);
  }
  else
  {
    own_cig = cigar_code(own_str.data(), own_str.size());
    own_end_val = (uint32_t) a_end;
  }
  if (tmp.begin_clips() != 0)
    own_bp = (uint32_t) a_start;
  else if (tmp.end_clips() != 0)
    own_bp = (uint32_t) a_end;
  else
    poison = true;
  if (sa_c.begin_clips() != 0)
    sa_bp = sa_start;
  else if (sa_c.end_clips() != 0)
    sa_bp = sa_end;
  else
    poison = true;
  uint64_t sa_cig = cigar_code(f[3].data(), f[3].size());
  if (!secondary)
  {
    t.prim_chr = own_chr;
    t.prim_start = (uint32_t) a_start;
    t.prim_end = own_end_val;
    t.prim_cigar = own_cig;
    t.prim_bp = own_bp;
    t.sec_chr = sa_chr;
    t.sec_start = sa_start;
    t.sec_end = sa_end;
    t.sec_cigar = sa_cig;
    t.sec_bp = sa_bp;
  }
  else
  {
    t.prim_chr = sa_chr;
    t.prim_start = sa_start;
    t.prim_end = sa_end;
    t.prim_cigar = sa_cig;
    t.prim_bp = sa_bp;
    t.sec_chr = own_chr;
    t.sec_start = (uint32_t) a_start;
    t.sec_end = own_end_val;
    t.sec_cigar = own_cig;
    t.sec_bp = own_bp;
  }
  if (poison) t.flags |= 2u;
  return true;
}

struct VoteIn
{
  uint64_t qhash;
  uint32_t secondary;
  int32_t prim_chr, sec_chr;
  uint32_t prim_start, prim_end, prim_bp, sec_start, sec_end, sec_bp;
  uint64_t prim_cigar, sec_cigar;
  uint32_t chr_check = 0;  // second hash of an SA contig name that is not a known name (bk_split.reserved)
};

// BreakID.cc:577-857 find_bp_pair ("update version"), bp_pos_error = 2
void find_bp_pair(const std::vector<VoteIn> &s1, const std::vector<VoteIn> &s2, int p1_chr, int32_t &p1_bp,
                  int32_t &p2_bp, int &num)
{
  std::vector<std::pair<int32_t, int32_t>> upd;
  // qname order of the outer std::map does not influence the result (counts and string-keyed map)
  for (auto &a : s1)
    for (auto &b : s2)
    {
      if (a.qhash != b.qhash) continue;
      bool c = (a.secondary != b.secondary) && a.prim_chr == b.prim_chr && a.sec_chr == b.sec_chr && a.chr_check == b.chr_check &&
               a.prim_start == b.prim_start && a.sec_start == b.sec_start && a.prim_end == b.prim_end &&
               a.sec_end == b.sec_end && a.prim_cigar == b.prim_cigar && a.sec_cigar == b.sec_cigar &&
               a.prim_bp == b.prim_bp && a.sec_bp == b.sec_bp;
      if (!c) continue;
      if (a.prim_chr == p1_chr)
        upd.emplace_back((int32_t) a.prim_bp, (int32_t) a.sec_bp);
      else
        upd.emplace_back((int32_t) a.sec_bp, (int32_t) a.prim_bp);
    }
  std::map<std::string, int> cnt;
  for (auto &u : upd) cnt[std::to_string(u.first) + "," + std::to_string(u.second)] = 0;
  for (auto &kv : cnt)
  {
    size_t comma = kv.first.find(',');
    uint32_t t1 = (uint32_t) std::stoull(kv.first.substr(0, comma));
    uint32_t t2 = (uint32_t) std::stoull(kv.first.substr(comma + 1));
    for (auto &u : upd)
    {
      // int32 vs uint32: the usual arithmetic conversions make all four comparisons unsigned (:820-821)
      if (((uint32_t) u.first <= t1 + 2u && (uint32_t) u.first >= t1 - 2u) &&
          ((uint32_t) u.second <= t2 + 2u && (uint32_t) u.second >= t2 - 2u))
        kv.second++;
    }
  }
  int best = 0;
  for (auto &kv : cnt)
    if (best < kv.second)
    {
      best = kv.second;
      size_t comma = kv.first.find(',');
      p1_bp = (int32_t) (uint32_t) std::stoull(kv.first.substr(0, comma));
      p2_bp = (int32_t) (uint32_t) std::stoull(kv.first.substr(comma + 1));
    }
  num = best;
}

}  // namespace

// =============================================================================================
extern "C" {

typedef struct Oracle ora;

ora *ora_new(const uint32_t *target_len, const char *const *target_name, int nt)
{
  Oracle *o = new Oracle();
  uint32_t acc = 0;
  for (int i = 0; i < nt; ++i)
  {
    o->tlen.push_back(target_len[i]);
    o->tname.push_back(target_name[i]);
    o->tprefix.push_back(acc);
    acc += target_len[i];
    if (!o->name_id.count(target_name[i])) o->name_id[target_name[i]] = i;
  }
  o->tprefix.push_back(acc);
  if (!o->name_id.count("")) o->name_id[""] = nt;
  if (!o->name_id.count("*")) o->name_id["*"] = nt + 1;
  for (int t = 0; t < 24; ++t)
  {
    std::string s = chrom_id_to_name(t);
    if (!o->name_id.count(s)) o->name_id[s] = nt + 2 + t;
  }
  return o;
}
void ora_free(ora *o) { delete o; }
const char *ora_last_error(ora *o) { return o->err.c_str(); }

int ora_set_records(ora *o, const bk_soa *s)
{
  o->s = *s;
  int mx = 1;
  for (uint64_t i = 0; i < s->n; ++i)
  {
    int sp = o->endpos(i) - s->pos[i];
    if (sp > mx) mx = sp;
  }
  o->maxspan = mx;
  return 0;
}

// A1: BreakID.cc:1909-1954
int ora_isize_stats(ora *o, double *mean, double *sd)
{
  const bk_soa &s = o->s;
  const uint32_t filter = 0x4 | 0x100 | 0x200 | 0x400;
  std::vector<int> v;
  for (uint64_t i = 0; i < s.n; ++i)
    if ((s.flag[i] & 1) && (s.flag[i] & 2) && !(s.flag[i] & filter)) v.push_back(abs(s.isize[i]));
  long total = 0, sd_total = 0;
  for (int x : v) total += x;
  double m = (double) total / (double) v.size();
  for (int x : v) sd_total += (((double) x) - m) * (((double) x) - m);
  o->mean = m;
  o->sd = std::sqrt(sd_total / (double) v.size());
  *mean = o->mean;
  *sd = o->sd;
  return 0;
}

// A2-A6: BreakID.cc:1362-1515
int ora_discordant_pairs(ora *o, int qual_i, double w)
{
  const bk_soa &s = o->s;
  long qual = qual_i;
  struct Buf
  {
    long flag, pos, mapq;
    int tid;
  };
  std::map<uint64_t, Buf> buffer;  // keyed by qname hash (the reference keys by the qname string)
  std::vector<bk_pair> all;
  for (uint64_t i = 0; i < s.n; ++i)
  {
    uint16_t f = s.flag[i];
    if (!((long) s.mapq[i] >= qual && !(f & 0x400) && !(f & 0x100) && (f & 1) && !(f & 2))) continue;
    long pos = (long) s.pos[i] + 1;
    auto it = buffer.find(s.qhash[i]);
    if (it != buffer.end())
    {
      if (o->rname(it->second.tid) != o->rname(s.tid[i]) || std::labs(pos - it->second.pos) >= w)
      {
        uint32_t c1 = o->gpos(s.tid[i], s.pos[i]);
        uint32_t c2 = o->gpos(s.mtid[i], s.mpos[i]);
        bk_pair p;
        memset(&p, 0, sizeof p);
        if (c1 <= c2)
        {
          p.p1_flag = f;
          p.p1_tid = s.tid[i];
          p.p1_pos = (uint32_t) pos;
          p.p1_mapq = s.mapq[i];
          p.x = c1;
          p.y = c2;
          p.p2_flag = (uint16_t) it->second.flag;
          p.p2_tid = it->second.tid;
          p.p2_pos = (uint32_t) it->second.pos;
          p.p2_mapq = (uint8_t) it->second.mapq;
        }
        else
        {
          p.p2_flag = f;
          p.p2_tid = s.tid[i];
          p.p2_pos = (uint32_t) pos;
          p.p2_mapq = s.mapq[i];
          p.x = c2;
          p.y = c1;
          p.p1_flag = (uint16_t) it->second.flag;
          p.p1_tid = it->second.tid;
          p.p1_pos = (uint32_t) it->second.pos;
          p.p1_mapq = (uint8_t) it->second.mapq;
        }
        p.p1_rev = (p.p1_flag & 0x10) ? 1 : 0;
        p.p2_rev = (p.p2_flag & 0x10) ? 1 : 0;
        p.rec = i;
        p.cluster = -1;
        all.push_back(p);
      }
      buffer.erase(it);
    }
    else
      buffer[s.qhash[i]] = Buf{(long) f, pos, (long) s.mapq[i], s.tid[i]};
  }
  std::map<std::string, std::vector<bk_pair>> groups;  // :1500-1512
  for (auto &p : all) groups[o->rname(p.p1_tid) + "_" + o->rname(p.p2_tid)].push_back(p);
  o->scan.clear();
  o->scan_off.assign(1, 0);
  o->group_keys.clear();
  uint32_t g = 0;
  for (auto &kv : groups)
  {
    uint32_t id = 0;
    for (auto p : kv.second)
    {
      p.group = g;
      p.id = id++;  // add_enspan_point_id :1287
      o->scan.push_back(p);
    }
    o->group_keys.push_back(kv.second[0].p1_tid);
    o->group_keys.push_back(kv.second[0].p2_tid);
    o->scan_off.push_back(o->scan.size());
    ++g;
  }
  return 0;
}

struct PX : bk_pair
{
  int k1 = 0, k2 = 0;
};

// A7-A9 for every group: BreakID.cc:119-137
int ora_mask_and_cluster(ora *o, double w, int fast)
{
  o->iso.clear();
  o->clustered.clear();
  o->iso_off.assign(1, 0);
  o->clustered_off.assign(1, 0);
  size_t ng = o->scan_off.size() - 1;
  for (size_t g = 0; g < ng; ++g)
  {
    std::vector<PX> v;
    for (uint64_t i = o->scan_off[g]; i < o->scan_off[g + 1]; ++i)
    {
      PX p;
      static_cast<bk_pair &>(p) = o->scan[i];
      v.push_back(p);
    }
    remove_isolated(v, w);
    for (auto &p : v) o->iso.push_back(p);
    o->iso_off.push_back(o->iso.size());
    if (v.size() >= 2)
    {
      if (fast)
        fast_cluster(v, w, 2);
      else
        ahc_cluster(v, w, 2);
      for (auto &p : v) o->clustered.push_back(p);
    }
    o->clustered_off.push_back(o->clustered.size());
  }
  return 0;
}

// per-record part of find_sa_reads for every record (the region loop selects from these later)
int ora_split_evidence(ora *o)
{
  o->splits.clear();
  for (uint64_t i = 0; i < o->s.n; ++i)
  {
    bk_split t;
    if (record_split(*o, i, t)) o->splits.push_back(t);
  }
  return 0;
}

// A10: BreakID.cc:225-352
int ora_cluster_summary(ora *o, double w)
{
  o->clusters.clear();
  size_t ng = o->clustered_off.size() - 1;
  for (size_t g = 0; g < ng; ++g)
  {
    std::map<long, std::vector<size_t>> idx;
    // the preceding std::sort by cluster (:144) only permutes members inside a cluster; every
    // quantity below is order independent (integer sums, min, max, set union)
    for (uint64_t i = o->clustered_off[g]; i < o->clustered_off[g + 1]; ++i) idx[o->clustered[i].cluster].push_back(i);
    for (auto &kv : idx)
    {
      bk_cluster c;
      memset(&c, 0, sizeof c);
      const bk_pair &p0 = o->clustered[kv.second[0]];
      c.group = (uint32_t) g;
      c.id = (int32_t) kv.first;
      c.p1_tid = p0.p1_tid;
      c.p2_tid = p0.p2_tid;
      uint64_t s1 = 0, s2 = 0;
      c.p1_min = c.p1_max = p0.p1_pos;
      c.p2_min = c.p2_max = p0.p2_pos;
      uint32_t type = 0;
      for (size_t i : kv.second)
      {
        const bk_pair &p = o->clustered[i];
        s1 += p.p1_pos;
        s2 += p.p2_pos;
        c.p1_min = std::min(c.p1_min, p.p1_pos);
        c.p1_max = std::max(c.p1_max, p.p1_pos);
        c.p2_min = std::min(c.p2_min, p.p2_pos);
        c.p2_max = std::max(c.p2_max, p.p2_pos);
        if (o->rname(p.p1_tid) != o->rname(p.p2_tid))
          type |= BK_TYPE_DIFF_CHR;
        else
        {
          if (p.p1_rev && !p.p2_rev) type |= BK_TYPE_ABS_REVERSE;
          if (p.p1_rev == p.p2_rev) type |= BK_TYPE_SAME_ORIENT;
          if (!p.p1_rev && p.p2_rev) type |= BK_TYPE_DEFAULT_ORIENT;
        }
      }
      c.n_drp = (uint32_t) kv.second.size();
      uint64_t m1 = (uint32_t) ((double) s1 / (double) c.n_drp);
      uint64_t m2 = (uint32_t) ((double) s2 / (double) c.n_drp);
      c.p1_mean = (uint32_t) m1;
      c.p2_mean = (uint32_t) m2;
      c.type_mask = type;
      c.p1_exact = (uint32_t) -1;
      c.p2_exact = -1;
      int64_t dist = (int64_t) (m1 - m2);
      bool same = o->rname(c.p1_tid) == o->rname(c.p2_tid);
      if (!(same && dist <= 2 * w && dist >= -2 * w))
      {
        c.flags = 1;
        o->clusters.push_back(c);
      }
    }
  }
  return 0;
}

namespace {
// region select with the htslib predicate (hts.c:1963-1965) over the coordinate-sorted table
void region_range(Oracle &o, int tid, int beg, int end, uint64_t &lo, uint64_t &hi)
{
  const bk_soa &s = o.s;
  // records are sorted by (tid, pos) with tid == -1 last
  auto key_lt = [&](uint64_t i, int t, long p) {
    uint32_t a = (uint32_t) s.tid[i], b = (uint32_t) t;  // -1 -> max
    if (a != b) return a < b;
    return (long) s.pos[i] < p;
  };
  auto lower = [&](int t, long p) {
    uint64_t a = 0, b = s.n;
    while (a < b)
    {
      uint64_t m = (a + b) / 2;
      if (key_lt(m, t, p)) a = m + 1; else b = m;
    }
    return a;
  };
  lo = lower(tid, (long) beg - o.maxspan);
  hi = lower(tid, (long) end);
}

// find_sa_reads, BreakID.cc:868-1037: region loop + region rejection rule
bool sa_region(Oracle &o, int tid, uint32_t rstart, uint32_t rend, std::vector<VoteIn> &out, bool &poison)
{
  out.clear();
  int beg = (int) rstart, end = (int) rend;  // uint32 -> int at bam_iter_query (:881)
  if (beg < 0) beg = 0;                       // hts.c:1776
  if (end < beg || tid < 0) return false;     // NULL iterator: unreachable for valid inputs
  uint64_t lo, hi;
  region_range(o, tid, beg, end, lo, hi);
  long cov = 0, ev = 0;
  for (uint64_t i = lo; i < hi; ++i)
  {
    if (o.s.tid[i] != tid || !(o.s.pos[i] < end && o.endpos(i) > beg)) continue;
    ++cov;
    bk_split t;
    if (record_split(o, i, t))
    {
      ++ev;
      if (t.flags & 2) poison = true;
      VoteIn v{t.qhash, t.flags & 1u, t.prim_chr, t.sec_chr, t.prim_start, t.prim_end, t.prim_bp,
               t.sec_start, t.sec_end, t.sec_bp, t.prim_cigar, t.sec_cigar, t.reserved};
      out.push_back(v);
    }
  }
  if (cov < 5 || ev < 2) out.clear();
  return !out.empty();
}

// cal_single_base_depth, util_bed.cc:154-192
uint32_t base_depth(Oracle &o, int tid, uint64_t pos)
{
  int beg = (int) (pos - 1), end = (int) pos;
  if (beg < 0) beg = 0;
  if (end < beg || tid < 0) return 0;
  uint64_t lo, hi;
  region_range(o, tid, beg, end, lo, hi);
  uint32_t d = 0;
  for (uint64_t i = lo; i < hi; ++i)
  {
    if (o.s.tid[i] != tid || !(o.s.pos[i] < end && o.endpos(i) > beg)) continue;
    if (o.s.mapq[i] > 0 && !(o.s.flag[i] & 0x400) && (o.s.flag[i] & 1)) ++d;
  }
  return d;
}
}  // namespace

// A11-A16: BreakID.cc:390-490
int ora_split_breakpoints(ora *o, double wd)
{
  const int w = (int) wd;  // :390 `const int w`
  std::vector<VoteIn> s1, s2;
  for (auto &c : o->clusters)  // every cluster stays in the stage output; flags bit1 marks the survivors (:481)
  {
    uint32_t r1s = (uint32_t) ((uint64_t) c.p1_mean - w), r1e = (uint32_t) ((uint64_t) c.p1_mean + w);
    uint32_t r2s = (uint32_t) ((uint64_t) c.p2_mean - w), r2e = (uint32_t) ((uint64_t) c.p2_mean + w);
    bool poison = false;
    bool ok1 = sa_region(*o, c.p1_tid, r1s, r1e, s1, poison);
    if (poison)
    {
      o->err = "error cigar";
      return BK_ERR_CIGAR;
    }
    bool ok2 = false;
    if (ok1) ok2 = sa_region(*o, c.p2_tid, r2s, r2e, s2, poison);
    if (poison)
    {
      o->err = "error cigar";
      return BK_ERR_CIGAR;
    }
    if (ok1 && ok2)
    {
      int32_t b1 = -1, b2 = -1;
      int num = 0;
      find_bp_pair(s1, s2, o->intern(o->rname(c.p1_tid)), b1, b2, num);
      if (num >= 2)
      {
        c.p1_exact = (uint32_t) b1;
        c.p2_exact = b2;
        c.n_sr = (uint32_t) num;
        c.depth1 = base_depth(*o, c.p1_tid, (uint64_t) c.p1_exact);
        c.depth2 = base_depth(*o, c.p2_tid, (uint64_t) (int64_t) c.p2_exact);
        c.flags |= 2;
      }
    }
  }
  return 0;
}

int ora_fetch(ora *o, int stage, const void **data, uint64_t *count, const uint64_t **group_off, uint32_t *n_groups)
{
  const uint64_t *off = nullptr;
  uint32_t ng = 0;
  switch (stage)
  {
  case BK_STAGE_SCAN: *data = o->scan.data(); *count = o->scan.size(); off = o->scan_off.data(); ng = (uint32_t) o->scan_off.size() - 1; break;
  case BK_STAGE_ISO: *data = o->iso.data(); *count = o->iso.size(); off = o->iso_off.data(); ng = (uint32_t) o->iso_off.size() - 1; break;
  case BK_STAGE_CLUSTERED: *data = o->clustered.data(); *count = o->clustered.size(); off = o->clustered_off.data(); ng = (uint32_t) o->clustered_off.size() - 1; break;
  case BK_STAGE_SPLITS: *data = o->splits.data(); *count = o->splits.size(); break;
  case BK_STAGE_CLUSTERS: *data = o->clusters.data(); *count = o->clusters.size(); break;
  case BK_STAGE_GROUP_KEYS: *data = o->group_keys.data(); *count = o->group_keys.size() / 2; break;
  default: return BK_ERR_ARG;
  }
  if (group_off) *group_off = off;
  if (n_groups) *n_groups = ng;
  return 0;
}

// ---- unit entry points (pinned against oracle/_ref/ref_units and ref_harness) ------------------
// cigar: kind 0 = text, 1 = BAM words.  out = {begin, end, reflen, nmatch, complementary}; rolled text in buf.
int ora_unit_cigar(int kind, const char *c1, const uint32_t *words, uint32_t nwords, const char *c2, int e,
                   int *out, char *buf, size_t buflen)
{
  Roller r;
  if (kind == 0) r.set_text(c1, strlen(c1)); else r.set_bam(words, nwords);
  std::string s = r.str();
  if (s.empty()) s = "*";
  snprintf(buf, buflen, "%s", s.c_str());
  out[0] = r.begin_clips();
  out[1] = r.end_clips();
  out[2] = r.reflen();
  out[3] = r.matches();
  out[4] = is_complementary(r, c2, strlen(c2), e) ? 1 : 0;
  return 0;
}

// ahc: returns num_nodes; nodes_out rows of 5 ints {idx,is_root,num_points,m0,m1}; points_out flattened
int ora_unit_ahc(const uint32_t *x, const uint32_t *y, uint32_t n, long T, int *nodes_out, int *points_out)
{
  Ahc c;
  std::vector<uint32_t> xs(x, x + n), ys(y, y + n);
  ahc_run(c, xs, ys, T);
  size_t pp = 0;
  for (size_t i = 0; i < c.nodes.size(); ++i)
  {
    nodes_out[i * 5 + 0] = (int) i;
    nodes_out[i * 5 + 1] = c.nodes[i].is_root;
    nodes_out[i * 5 + 2] = c.nodes[i].num_points;
    nodes_out[i * 5 + 3] = c.nodes[i].m0;
    nodes_out[i * 5 + 4] = c.nodes[i].m1;
    for (int p : c.nodes[i].points) points_out[pp++] = p;
  }
  return (int) c.nodes.size();
}

// mode 0 = mask(distance=(long)w), 1 = remove_isolated(w), 2 = fast(w).  ids_out/cluster_out sized n.
int ora_unit_points(int mode, const uint32_t *x, const uint32_t *y, uint32_t n, double w, uint32_t *ids_out,
                    int32_t *cluster_out, int *k_out)
{
  std::vector<XY> v(n);
  for (uint32_t i = 0; i < n; ++i) v[i] = XY{x[i], y[i], i, -1, 0, 0};
  int k = 0;
  if (mode == 0) mask_pairs(v, (long) w);
  else if (mode == 1) remove_isolated(v, w);
  else k = fast_cluster(v, w, 2);
  for (size_t i = 0; i < v.size(); ++i)
  {
    ids_out[i] = v[i].id;
    cluster_out[i] = v[i].cluster;
  }
  if (k_out) *k_out = k;
  return (int) v.size();
}

// vote: rows of bk_split-like VoteIn for both sides
int ora_unit_vote(const bk_split *s1, uint32_t n1, const bk_split *s2, uint32_t n2, int p1_chr, int32_t *out3)
{
  std::vector<VoteIn> a, b;
  for (uint32_t i = 0; i < n1; ++i)
    a.push_back(VoteIn{s1[i].qhash, s1[i].flags & 1u, s1[i].prim_chr, s1[i].sec_chr, s1[i].prim_start, s1[i].prim_end,
                       s1[i].prim_bp, s1[i].sec_start, s1[i].sec_end, s1[i].sec_bp, s1[i].prim_cigar, s1[i].sec_cigar, s1[i].reserved});
  for (uint32_t i = 0; i < n2; ++i)
    b.push_back(VoteIn{s2[i].qhash, s2[i].flags & 1u, s2[i].prim_chr, s2[i].sec_chr, s2[i].prim_start, s2[i].prim_end,
                       s2[i].prim_bp, s2[i].sec_start, s2[i].sec_end, s2[i].sec_bp, s2[i].prim_cigar, s2[i].sec_cigar, s2[i].reserved});
  int32_t b1 = -1, b2 = -1;
  int num = 0;
  find_bp_pair(a, b, p1_chr, b1, b2, num);
  out3[0] = b1;
  out3[1] = b2;
  out3[2] = num;
  return 0;
}

// std::sort of (key, original index) records by key, per group: the permutation the product must reproduce
int ora_unit_std_sort(const uint32_t *key, const uint64_t *group_off, uint32_t n_groups, uint32_t *perm_out)
{
  struct KI { uint32_t key, id; };
  for (uint32_t g = 0; g < n_groups; ++g)
  {
    std::vector<KI> v;
    for (uint64_t p = group_off[g]; p < group_off[g + 1]; ++p) v.push_back(KI{key[p], (uint32_t) p});
    std::sort(v.begin(), v.end(), [](KI a, KI b) { return a.key < b.key; });
    for (size_t i = 0; i < v.size(); ++i) perm_out[group_off[g] + i] = v[i].id;
  }
  return 0;
}

// find_sa_reads on one region (BreakID.cc:868-1037): tuples after the region verdict, as bk_split rows
int ora_unit_region(ora *o, int tid, uint32_t start, uint32_t end, bk_split *out, uint32_t cap)
{
  std::vector<VoteIn> v;
  bool poison = false;
  sa_region(*o, tid, start, end, v, poison);
  uint32_t n = 0;
  for (auto &t : v)
  {
    if (n >= cap) break;
    bk_split s;
    memset(&s, 0, sizeof s);
    s.qhash = t.qhash;
    s.flags = t.secondary;
    s.prim_chr = t.prim_chr; s.sec_chr = t.sec_chr;
    s.prim_start = t.prim_start; s.prim_end = t.prim_end; s.prim_bp = t.prim_bp;
    s.sec_start = t.sec_start; s.sec_end = t.sec_end; s.sec_bp = t.sec_bp;
    s.prim_cigar = t.prim_cigar; s.sec_cigar = t.sec_cigar;
    out[n++] = s;
  }
  return (int) v.size();
}
uint32_t ora_unit_depth(ora *o, int tid, uint64_t pos) { return base_depth(*o, tid, pos); }

// diagnostic: enable the heapsort-branch probe / read and reset its counters {sorts, heap segments, elements in them, largest}
void ora_sort_probe(int on) { g_sort_probe = on != 0; g_sort_stats = SortStats(); }
void ora_sort_stats(uint64_t out[4])
{
  out[0] = g_sort_stats.sorts; out[1] = g_sort_stats.heap_segments; out[2] = g_sort_stats.heap_elems; out[3] = g_sort_stats.max_heap;
}

uint64_t ora_text_hash(const char *s, size_t len) { return cigar_code(s, len); }  // code of a CIGAR text (bk_split.prim_cigar / sec_cigar)
int ora_name_id(ora *o, const char *name) { return o->intern(name); }

}  // extern "C"
