// TEST INFRASTRUCTURE ONLY.  Stage-dump harness over the reference's own functions in
// /root/reference/src/BreakID.cc (compiled from where it lies; main() renamed).  It replays the
// body of the reference's main() (BreakID.cc:93-167) and writes every intermediate the product's
// C-ABI stages must reproduce.  Output of this tool is committed as golden vectors (tests/golden/),
// the tool itself and the reference never travel to the GPU box.
//
//   ref_harness stages <bam> <nib_dir> <qual> <fast 0|1> <out.txt> [nocall]   (nocall: stop after clustering, no region queries)
//   ref_harness sa     <bam> <chr> <start> <end>            (find_sa_reads dump, BreakID.cc:868)
//   ref_harness depth  <bam> <chr> <pos>                    (cal_single_base_depth, util_bed.cc:154)
//   ref_harness mask   <distance>        stdin: n, then "x y" lines  (mask_pairs_chr_pos, :1813)
//   ref_harness iso    <w>               stdin: n, then "x y" lines  (remove_isolated_pairs, :1271)
//   ref_harness fast   <w>               stdin: n, then "x y" lines  (find_cluster_pairs_enspan_fast, :1046)
//   ref_harness vote   stdin: side-1 and side-2 tuple tables          (find_bp_pair, :577)
#include <cstdlib>
#include <cstdio>
#include "/root/reference/src/BreakID.h"
static inline const char *oracle_installdir()
{
  const char *e = getenv("BREAKID_REF_INSTALLDIR");
  return e ? e : "/nonexistent";
}
#undef INSTALLDIR
#define INSTALLDIR oracle_installdir()
#define main reference_main
#include "/root/reference/src/BreakID.cc"
#undef main

static void dump_pairs(FILE *f, const char *tag, const string &key, vector<discordant_pair> &v, bool with_cluster)
{
  fprintf(f, "%s %s %zu\n", tag, key.c_str(), v.size());
  for (auto &p : v)
  {
    fprintf(f, "%s %ld %ld %s %s %u %u %ld %ld %c %c %u %u", p.id.c_str(), p.p1_flag, p.p2_flag, p.p1_chr.c_str(),
            p.p2_chr.c_str(), p.p1_pos, p.p2_pos, p.p1_mapq, p.p2_mapq, p.p1_strand, p.p2_strand, p.p1_chr_pos,
            p.p2_chr_pos);
    if (with_cluster) fprintf(f, " %d", p.cluster);
    fprintf(f, " %s\n", p.qname.c_str());
  }
}

static int run_stages(int argc, char **argv)
{
  string bam = argv[2], nib = argv[3];
  long qual = atol(argv[4]);
  bool fast = atoi(argv[5]) != 0;
  FILE *f = fopen(argv[6], "w");
  if (!f) return 2;
  const bool nocall = argc > 7 && string(argv[7]) == "nocall";
  stringstream sink;
  streambuf *old = cout.rdbuf(sink.rdbuf());

  vector<double> insert;
  get_mean_insert_size(bam, insert);
  int times = 2;
  double w = times * sqrt(times) * (insert[0] + 3 * insert[1]);
  fprintf(f, "insert %a %a %a\n", insert[0], insert[1], w);
  map<string, vector<discordant_pair>> enspan_map;
  scan_discordant_pairs(bam, "hg19", qual, w, enspan_map, nib);
  vector<cluster_info> cluster, tmp_cluster_vec;
  for (auto &chr_it : enspan_map)
  {
    add_enspan_point_id(chr_it.second);
    dump_pairs(f, "scan", chr_it.first, chr_it.second, false);
    remove_isolated_pairs(chr_it.second, w);
    dump_pairs(f, "iso", chr_it.first, chr_it.second, false);
    if (chr_it.second.size() >= 2)
    {
      int roots;
      if (fast)
        roots = find_cluster_pairs_enspan_fast(chr_it.second, w, 2);
      else
        roots = find_cluster_pairs_enspan_ahc(chr_it.second, w, 1, 2);
      fprintf(f, "roots %d\n", roots);
      dump_pairs(f, "clustered", chr_it.first, chr_it.second, true);
      if (nocall) continue;
      sort(chr_it.second.begin(), chr_it.second.end(), cmp_enspan_id);
      vector<bam1_t *> split_reads;
      findClusterBreakPointInfoSaTag(bam, chr_it.second, w, tmp_cluster_vec, split_reads, nib);
      fprintf(f, "clusters %s %zu\n", chr_it.first.c_str(), tmp_cluster_vec.size());
      for (auto &c : tmp_cluster_vec)
      {
        fprintf(f, "%ld %s %s %lu %lu %u %u %u %u %u %d %ld %ld %s %a %a %a %a %d\n", c.id, c.p1_chr.c_str(),
                c.p2_chr.c_str(), (unsigned long) c.p1_mean_pos, (unsigned long) c.p2_mean_pos, c.p1_min_pos,
                c.p1_max_pos, c.p2_min_pos, c.p2_max_pos, c.p1_exact_pos, c.p2_exact_pos, c.n_discordant_pair,
                c.n_split_read, c.fusion_type.c_str(), c.p1_bp_depth, c.p2_bp_depth, (double) c.p1_alle_freq,
                (double) c.p2_alle_freq, c.is_rpt ? 1 : 0);
        cluster.push_back(c);
      }
      tmp_cluster_vec.clear();
    }
  }
  cout.rdbuf(old);
  fclose(f);
  return 0;
}

static int run_sa(char **argv)
{
  samfile_t *fp = samopen(argv[2], "rb", 0);
  bam_index_t *idx = bam_index_load(argv[2]);
  if (!fp || !idx) return 2;
  map<string, vector<split_align_pair>> m;
  find_sa_reads(fp, argv[3], (uint32_t) strtoul(argv[4], 0, 10), (uint32_t) strtoul(argv[5], 0, 10), m, idx);
  size_t n = 0;
  for (auto &kv : m) n += kv.second.size();
  printf("tuples %zu\n", n);
  for (auto &kv : m)
    for (auto &t : kv.second)
      printf("%s %d %d %s %u %u %s %u %s %u %u %s %u\n", t.read_name.c_str(), t.flag, t.secondary ? 1 : 0,
             t.primary_chr.c_str(), t.primary_start, t.primary_end, t.primary_cigar_str.c_str(), t.primary_bp,
             t.secondary_chr.c_str(), t.secondary_start, t.secondary_end, t.secondary_cigar_str.c_str(),
             t.secondary_bp);
  return 0;
}

static int run_depth(char **argv)
{
  samfile_t *fp = samopen(argv[2], "rb", 0);
  bam_index_t *idx = bam_index_load(argv[2]);
  if (!fp || !idx) return 2;
  double d = cal_single_base_depth(argv[3], strtoull(argv[4], 0, 10), fp, idx);
  printf("%a\n", d);
  return 0;
}

static void read_points(vector<discordant_pair> &v)
{
  size_t n;
  cin >> n;
  v.resize(n);
  for (size_t i = 0; i < n; ++i)
  {
    unsigned long x, y;
    cin >> x >> y;
    v[i].p1_chr_pos = (uint32_t) x;
    v[i].p2_chr_pos = (uint32_t) y;
    v[i].id = to_string(i);
    v[i].cluster = -1;
  }
}

static void print_points(vector<discordant_pair> &v, bool with_cluster)
{
  printf("n %zu\n", v.size());
  for (auto &p : v)
  {
    if (with_cluster)
      printf("%s %d\n", p.id.c_str(), p.cluster);
    else
      printf("%s\n", p.id.c_str());
  }
}

static int run_vote()
{
  // input: "<n1>" then n1 lines, "<n2>" then n2 lines, then "<p1_chr> <p2_chr>"; line =
  // qname secondary prim_chr prim_start prim_end prim_cigar prim_bp sec_chr sec_start sec_end sec_cigar sec_bp
  map<string, vector<split_align_pair>> side[2];
  bam1_t *dummy = bam_init1();
  for (int s = 0; s < 2; ++s)
  {
    size_t n;
    cin >> n;
    for (size_t i = 0; i < n; ++i)
    {
      split_align_pair t;
      int sec;
      cin >> t.read_name >> sec >> t.primary_chr >> t.primary_start >> t.primary_end >> t.primary_cigar_str >>
        t.primary_bp >> t.secondary_chr >> t.secondary_start >> t.secondary_end >> t.secondary_cigar_str >>
        t.secondary_bp;
      t.secondary = sec != 0;
      t.current_align = dummy;
      side[s][t.read_name].push_back(t);
    }
  }
  string c1, c2;
  cin >> c1 >> c2;
  breakpoint_pair bp;
  vector<bam1_t *> split_reads;
  find_bp_pair(side[0], side[1], bp, c1, c2, split_reads, 2);
  printf("%d %d %d\n", bp.p1_bp, bp.p2_bp, bp.encompass_num);
  return 0;
}

int main(int argc, char **argv)
{
  string cmd = argc > 1 ? argv[1] : "";
  if (cmd == "stages" && (argc == 7 || argc == 8)) return run_stages(argc, argv);
  if (cmd == "sa" && argc == 6) return run_sa(argv);
  if (cmd == "depth" && argc == 5) return run_depth(argv);
  if (cmd == "mask" && argc == 3)
  {
    vector<discordant_pair> v;
    read_points(v);
    mask_pairs_chr_pos(v, atol(argv[2]));
    print_points(v, false);
    return 0;
  }
  if (cmd == "iso" && argc == 3)
  {
    vector<discordant_pair> v;
    read_points(v);
    remove_isolated_pairs(v, atof(argv[2]));
    print_points(v, false);
    return 0;
  }
  if (cmd == "fast" && argc == 3)
  {
    vector<discordant_pair> v;
    read_points(v);
    stringstream sink;
    int k = v.empty() ? 0 : find_cluster_pairs_enspan_fast(v, atof(argv[2]), 2);
    printf("k %d\n", k);
    print_points(v, true);
    return 0;
  }
  if (cmd == "vote") return run_vote();
  fprintf(stderr, "usage: see header of oracle/harness/ref_harness.cc\n");
  return 2;
}
