// TEST INFRASTRUCTURE ONLY.  Unit harness over the reference's own util_cluster.cc and
// CigarRoller.cc / Cigar.cc (linked from /root/reference/src, never copied).  Produces the golden
// vectors committed under tests/golden/ (see tools/make_golden.py).
//
//   ref_units ahc <T>      stdin: "n" then n lines "x y"  (uint32 coordinates)
//                          stdout: "nodes <num_nodes>" then one line per node:
//                                  idx is_root num_points m0 m1 | p0 p1 ...   (m = merged children or -1)
//   ref_units cigar        stdin: lines "<kind> <c1> <c2> <e>"; kind t = c1 is text, b = c1 is
//                                  comma separated BAM words (len<<4|op)
//                          stdout: rolled-string begin end reflen nmatch complementary
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include "/root/reference/src/util_cluster.h"
#include "/root/reference/src/CigarRoller.h"

static int run_ahc(long T)
{
  size_t n;
  if (!(std::cin >> n)) return 2;
  std::vector<point> pts(n);
  for (size_t i = 0; i < n; ++i)
  {
    unsigned long x, y;
    std::cin >> x >> y;
    pts[i].pos.x = (uint32_t) x;
    pts[i].pos.y = (uint32_t) y;
    pts[i].label = "p";
  }
  cluster_struct c;
  // the reference prints timing lines on stdout; park them on stderr-less buffer
  std::stringstream sink;
  std::streambuf *old = std::cout.rdbuf(sink.rdbuf());
  init_cluster(c, T, pts, 1);
  std::cout.rdbuf(old);
  printf("nodes %d\n", c.num_nodes);
  for (int i = 0; i < c.num_nodes; ++i)
  {
    node &nd = c.nodes[i];
    int m0 = nd.merged.size() > 0 ? nd.merged[0] : -1;
    int m1 = nd.merged.size() > 1 ? nd.merged[1] : -1;
    printf("%d %d %d %d %d |", i, nd.is_root, nd.num_points, m0, m1);
    for (int j = 0; j < nd.num_points; ++j) printf(" %d", nd.points[j]);
    printf("\n");
  }
  return 0;
}

static int run_cigar()
{
  std::string kind, c1, c2;
  int e;
  while (std::cin >> kind >> c1 >> c2 >> e)
  {
    CigarRoller r;
    if (kind == "t")
      r.Set(c1.c_str());
    else
    {
      std::vector<uint32_t> words;
      std::stringstream ss(c1);
      std::string tok;
      while (std::getline(ss, tok, ',')) words.push_back((uint32_t) strtoul(tok.c_str(), NULL, 10));
      r.Set(words.data(), (uint16_t) words.size());
    }
    std::string rolled;
    r.getCigarString(rolled);
    if (rolled.empty()) rolled = "*";
    bool comp = r.is_complementary_cigar(c2, e);
    printf("%s %d %d %d %d %d\n", rolled.c_str(), r.getNumBeginClips(), r.getNumEndClips(),
           r.getExpectedReferenceBaseCount(), r.getNumMatches(), comp ? 1 : 0);
  }
  return 0;
}

int main(int argc, char **argv)
{
  if (argc >= 3 && std::string(argv[1]) == "ahc") return run_ahc(atol(argv[2]));
  if (argc >= 2 && std::string(argv[1]) == "cigar") return run_cigar();
  fprintf(stderr, "usage: ref_units ahc <T> | cigar\n");
  return 2;
}
