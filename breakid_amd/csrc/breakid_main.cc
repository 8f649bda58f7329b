// BreakID command line on top of libbreakid_hip.so: same options, same output files as the reference
// (src/BreakID.cc:6-192, help text src/BreakID.h:27-36).  The hot path (BreakID.cc:98-167 minus annotation)
// runs on the MI355X through the C ABI; this file is the host side the reference keeps in main():
// argument parsing, BAM decode into the columnar table, refGene/nib annotation (BreakID.cc:492-567,
// :1528-1793, RefSeqTranscript.cc, nibtools.cc, util_bam.cc:78-122, util_bed.cc:224-261) and the writers
// (:1170-1263).  There is no CPU implementation of the hot path in here.
#include <getopt.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../include/breakid_hip.h"
#include "../../include/breakid_multi.h"

// bam_index_load (htslib-1.3.1 sam.h:302 -> hts.c:2042 hts_idx_load, :1580 hts_idx_load_local, :1528 hts_idx_load_core): the index is
// <bam>.csi, <bam with its extension replaced>.csi, <bam>.bai, <...>.bai - the first that can be opened - and it must parse to the
// end: magic, counts, every bin's chunk list and every linear index (a truncated or foreign file gives NULL, i.e. the reference's
// "please index bam-file first" exit, BreakID.cc:411-416).  The hot path streams the whole file and does not use the offsets; what is
// reproduced here is which files the reference accepts.  gzread() reads plain and (B)GZF-compressed files alike, as bgzf_read does.
static bool index_loads(const std::string &bam)
{
  auto candidate = [&](const char *ext) -> std::string {
    std::string a = bam + ext;
    if (FILE *f = fopen(a.c_str(), "rb"))
    {
      fclose(f);
      return a;
    }
    size_t i = bam.size();
    while (i > 1 && bam[i - 1] != '.') --i;  // hts_idx_getfn: the last '.' at an index > 0
    if (i > 1)
    {
      a = bam.substr(0, i - 1) + ext;
      if (FILE *f = fopen(a.c_str(), "rb"))
      {
        fclose(f);
        return a;
      }
    }
    return std::string();
  };
  std::string fn = candidate(".csi");
  if (fn.empty()) fn = candidate(".bai");
  if (fn.empty()) return false;
  gzFile fp = gzopen(fn.c_str(), "rb");
  if (!fp) return false;
  auto rd = [&](void *dst, size_t n) { return n == 0 || gzread(fp, dst, (unsigned) n) == (int) n; };
  auto skip = [&](uint64_t n) {
    char buf[65536];
    while (n)
    {
      const size_t k = n < sizeof buf ? (size_t) n : sizeof buf;
      if (!rd(buf, k)) return false;
      n -= k;
    }
    return true;
  };
  bool ok = false;
  do
  {
    uint8_t magic[4];
    if (!rd(magic, 4)) break;
    int fmt;  // 0 CSI, 1 BAI, 2 TBI
    int32_t n_ref = 0;
    if (!memcmp(magic, "CSI\1", 4))
    {
      uint32_t x[3];
      if (!rd(x, 12) || !skip(x[2]) || !rd(&n_ref, 4)) break;
      fmt = 0;
    }
    else if (!memcmp(magic, "TBI\1", 4))
    {
      uint32_t x[8];
      if (!rd(x, 32) || !skip(x[7])) break;
      n_ref = (int32_t) x[0];
      fmt = 2;
    }
    else if (!memcmp(magic, "BAI\1", 4))
    {
      if (!rd(&n_ref, 4)) break;
      fmt = 1;
    }
    else
      break;
    bool good = true;
    for (int32_t i = 0; i < n_ref && good; ++i)
    {
      int32_t n_bin;
      if (!rd(&n_bin, 4))
      {
        good = false;
        break;
      }
      std::set<uint32_t> seen;
      for (int32_t j = 0; j < n_bin && good; ++j)
      {
        uint32_t key;
        int32_t n_chunk;
        uint64_t loff;
        good = rd(&key, 4) && seen.insert(key).second && (fmt != 0 || rd(&loff, 8)) && rd(&n_chunk, 4) && n_chunk >= 0 && skip((uint64_t) n_chunk << 4);
      }
      if (good && fmt != 0)
      {
        int32_t n_intv;
        good = rd(&n_intv, 4) && n_intv >= 0 && skip((uint64_t) n_intv << 3);
      }
    }
    ok = good;  // (the trailing n_no_coor is optional: hts.c:1575)
  } while (false);
  gzclose(fp);
  return ok;
}

#ifndef BREAKID_INSTALLDIR
#define BREAKID_INSTALLDIR "."
#endif

using std::string;
using std::vector;

static const char *HELP =
    " Usage: \n \t BreakID -i input.bam -o prefix -n nib_folder <options> \n\n \
     DESCRIPTION\n \
     \t -h -? -help \t help\n \
     \t -i*        \t input bam-file\n \
     \t -o*        \t output file (prefix only)\n \
     \t -n*        \t folder name to nib files\n \
     \t -q         \t encompassing reads quality thresholds  [20]\n\
     \t -t         \t distance relative to (sqrt(2)*(insert size mean +3* insert size sd))  [2]\n \
     \t -fast      \t use the fast cluster strategy [default no] \n \
     \t -all       \t no filter enspan out [default is filter]  \n ";

// ---- RefSeqTranscript.{h,cc} -------------------------------------------------------------------------------
struct Txpt
{
  string transcriptID, chrom, strand, geneName;
  uint32_t txStart = 0, txEnd = 0, cdsStart = 0, cdsEnd = 0, exonCount = 0, cDNALength = 0;
  vector<uint32_t> exonStarts, exonEnds, codingStarts, codingEnds, codingParts;
  int codingExonCount = 0;
};

static vector<uint32_t> split_to_int(const string &s)  // splitStringToInt(s, ","), empty tokens dropped
{
  vector<uint32_t> out;
  size_t st = 0;
  while (true)
  {
    size_t e = s.find(',', st);
    string tok = s.substr(st, e == string::npos ? string::npos : e - st);
    if (!tok.empty()) out.push_back((uint32_t) atol(tok.c_str()));
    if (e == string::npos) break;
    st = e + 1;
  }
  return out;
}

static Txpt parse_refgene_line(const string &line)  // RefSeqTranscript.cc:19-81 + removeUTR :94-142
{
  Txpt t;
  std::stringstream ss(line);
  string f[16];
  for (int i = 0; i < 16; ++i)
    if (!getline(ss, f[i], '\t')) f[i] = i ? f[i - 1] : "";  // getline leaves `tmp` unchanged at EOF
  t.transcriptID = f[1];
  t.chrom = f[2];
  t.strand = f[3];
  t.txStart = (uint32_t) atol(f[4].c_str());
  t.txEnd = (uint32_t) atol(f[5].c_str());
  t.cdsStart = (uint32_t) atol(f[6].c_str());
  t.cdsEnd = (uint32_t) atol(f[7].c_str());
  t.exonCount = (uint32_t) atol(f[8].c_str());
  t.exonStarts = split_to_int(f[9]);
  t.exonEnds = split_to_int(f[10]);
  t.geneName = f[12];
  if (t.cdsStart != t.cdsEnd)
  {
    for (uint32_t i = 0; i < t.exonCount && i < t.exonStarts.size() && i < t.exonEnds.size(); ++i)
    {
      uint32_t s = t.exonStarts[i], e = t.exonEnds[i];
      if (s < t.cdsEnd && e > t.cdsStart)
      {
        if (s < t.cdsStart && e > t.cdsStart && e <= t.cdsEnd) { t.codingStarts.push_back(t.cdsStart); t.codingEnds.push_back(e); }
        else if (s < t.cdsEnd && e > t.cdsEnd && s >= t.cdsStart) { t.codingStarts.push_back(s); t.codingEnds.push_back(t.cdsEnd); }
        else if (e > t.cdsEnd && s < t.cdsStart) { t.codingStarts.push_back(t.cdsStart); t.codingEnds.push_back(t.cdsEnd); }
        else { t.codingStarts.push_back(s); t.codingEnds.push_back(e); }
      }
    }
    t.codingExonCount = (int) t.codingStarts.size();
    for (size_t i = 0; i < t.codingStarts.size(); ++i) t.cDNALength += t.codingEnds[i] - t.codingStarts[i];
  }
  for (size_t i = 0; i < t.codingStarts.size(); ++i)  // add_cds_parts
  {
    t.codingParts.push_back(t.codingStarts[i]);
    t.codingParts.push_back(t.codingEnds[i]);
  }
  return t;
}

static bool read_refgene(const string &fn, vector<Txpt> &out)  // readRefSeqTranscript: NR_ transcripts skipped
{
  std::ifstream in(fn);
  if (!in.is_open()) return false;
  string line;
  while (getline(in, line, '\n'))
  {
    std::stringstream l2(line);
    string a, b;
    getline(l2, a, '\t');
    if (!getline(l2, b, '\t')) b = a;
    if (b.find("NR_") != string::npos) continue;
    out.push_back(parse_refgene_line(line));
  }
  return true;
}

// add_exon_num_anno, BreakID.cc:1753-1793
static void exon_numbers(const Txpt &t, long pos, int &s_no, int &e_no)
{
  s_no = e_no = 0;
  for (size_t i = 0; i + 1 < t.codingParts.size(); ++i)
  {
    if (pos >= (long) t.codingParts[i] && pos <= (long) t.codingParts[i + 1])
    {
      int idx = (int) i / 2 + 1;
      if (t.strand == "+")
      {
        s_no = idx;
        e_no = (i % 2 == 1) ? idx + 1 : idx;
      }
      if (t.strand == "-")
      {
        s_no = t.codingExonCount + 1 - (idx + 1);
        e_no = (i % 2 == 1) ? t.codingExonCount + 1 - idx : t.codingExonCount + 1 - (idx + 1);
      }
      break;
    }
  }
}

// one side of add_exon_anno, BreakID.cc:1549-1585 (find_the_longest_cds_txpt never updates its maximum, so the
// LAST overlapping transcript with cDNA > 0 wins, RefSeqTranscript.cc:311-320)
static void annotate_side(const vector<Txpt> &txpts, const string &chr, long pos, string &gene, string &exon_info, string &strand)
{
  if (pos == -1)
  {
    exon_info = gene = strand = ".";
    return;
  }
  vector<const Txpt *> hit;
  for (auto &t : txpts)
    if (chr == t.chrom && pos >= (long) t.txStart && pos <= (long) t.txEnd) hit.push_back(&t);
  if (hit.empty())
  {
    exon_info = ".";
    gene = "intergenic";
    strand = ".";
    return;
  }
  Txpt chosen;
  for (auto *t : hit)
    if ((int) t->cDNALength > 0) chosen = *t;
  gene = chosen.geneName;
  strand = chosen.strand;
  int a, b;
  exon_numbers(chosen, pos, a, b);
  exon_info = chosen.transcriptID + ":" + std::to_string(a) + "-" + std::to_string(b);
}

// ---- nib access: nibtools.cc:7-58, util_bam.cc:78-122 -------------------------------------------------------------------
struct Nib
{
  std::ifstream in;
  unsigned long nBases = 0;
  bool ok = false;
  void open(const string &fn)
  {
    in.open(fn, std::ios::binary);
    if (!in.is_open()) return;
    unsigned char raw[8];
    in.read((char *) raw, 8);
    unsigned long sig = raw[0] | (raw[1] << 8) | (raw[2] << 16) | ((unsigned long) raw[3] << 24);
    nBases = raw[4] | (raw[5] << 8) | (raw[6] << 16) | ((unsigned long) raw[7] << 24);
    ok = sig == 0x6be93d3aUL;
  }
  void base(char *out, unsigned long pos)  // leaves *out untouched on any failure, like the reference
  {
    if (!ok || pos >= nBases) return;
    in.seekg(8 + pos / 2);
    char r;
    in.read(&r, 1);
    int v = (pos % 2 == 0) ? ((r & 0xff) >> 4) : (r & 0x0f);
    static const char tab[16] = {'T', 'C', 'A', 'G', 'N', 'N', 'N', 'N', 'T', 'C', 'A', 'G', 'N', 'N', 'N', 'N'};
    *out = tab[v & 15];
  }
};

static string neighbour_seq(const string &nib_dir, const string &chr, int32_t bp)
{
  // left 20 (1-based bp-20 .. bp-1) + right 21 (bp .. bp+20), BreakID.cc:554-559
  Nib n;
  n.open(nib_dir + "/hg19_" + chr + ".nib");
  string s;
  char b = 'N';
  for (int32_t i = bp - 20; i < bp; ++i)
  {
    n.base(&b, (unsigned long) (long) (i - 1));
    s += b;
  }
  for (int32_t i = bp - 1; i < bp - 1 + 21; ++i)
  {
    n.base(&b, (unsigned long) (long) i);
    s += b;
  }
  return s;
}

static int longest_run(const string &s)  // find_longest_repeat_substring, util_bed.cc:224-261
{
  int best = 0;
  size_t i = 0;
  while (i < s.size())
  {
    size_t j = i + 1;
    while (j < s.size() && s[j] == s[i]) ++j;
    best = std::max(best, (int) (j - i));
    i = j;
  }
  return best;
}

static const char *fusion_type(uint32_t mask)  // determine_fusion_type_from_drp, BreakID.cc:1888-1907
{
  if (mask & BK_TYPE_DEFAULT_ORIENT) return "Deletion";
  if (mask & BK_TYPE_ABS_REVERSE) return "Duplication";
  if (mask & BK_TYPE_SAME_ORIENT) return "Inversion";
  if (mask & BK_TYPE_DIFF_CHR) return "Translocation";
  return "Unknown";
}

struct OutRow
{
  bk_cluster c;
  string p1_chr, p2_chr, g1, g2, e1, e2, s1, s2, rpt1, rpt2;
  bool is_rpt;
  float af1, af2;
};
static bool cmp_cluster(OutRow a, OutRow b) { return a.c.n_drp > b.c.n_drp; }  // BreakID.h:185-188 (by value, like the reference)

static void write_row(std::ostream &o, const OutRow &r)
{
  o << fusion_type(r.c.type_mask) << "\t";
  o << r.p1_chr << ":" << r.c.p1_exact << "\t";
  o << r.p2_chr << ":" << r.c.p2_exact << "\t";
  o << r.g1 << "\t" << r.s1 << ":" << r.e1 << "\t";
  o << r.g2 << "\t" << r.s2 << ":" << r.e2 << "\t";
  o << (long) r.c.n_drp << "\t" << (long) r.c.n_sr << "\t";
  o << (double) r.c.depth1 << "\t" << (double) r.c.depth2 << "\t";
  o << r.af1 << "\t" << r.af2 << "\t";
  o << r.rpt1 << "\t" << r.rpt2 << "\n";
}

static const char *HEADER =
    "Fusion_Type\tBreakPoint1\tBreakPoint2\tGene1\tBreakPoint_Info_Pair1\tGene2\tBreakPoint_Info_Pair2\tN_DRP\tN_SR\t"
    "BreakPoint1_Depth\tBreakPoint2_Depth\tBreakPoint1_AF\tBreakPoint2_AF\tBP1_Neighbour_Seq\tBP2_Neighbour_Seq\n";

int main(int argc, char *argv[])
{
  clock_t start = clock();
  // two lanes of chromosome-pair groups (csrc/api.hip: bk_mask_and_cluster) need more hardware queues than ROCm's default 4;
  // both are read when the runtime starts / at the first stage call, so they are set before anything touches the GPU
  setenv("GPU_MAX_HW_QUEUES", "20", 0);
  static struct option longopts[] = {{"help", 0, 0, 'h'}, {"i", 1, 0, 1}, {"o", 1, 0, 2}, {"q", 1, 0, 3}, {"n", 1, 0, 4},
                                     {"fast", 0, 0, 5},   {"t", 0, 0, 6}, {"all", 0, 0, 7}, {"gpu", 1, 0, 8}, {"gpus", 1, 0, 9},
                                     {"comm", 1, 0, 10},  {0, 0, 0, 0}};
  string inp_file, out_file, nib_dir, build = "hg19";
  int qual = 20, device = 0, n_gpus = 0, transport = BK_TRANSPORT_AUTO;  // -gpus N: one sample over N GPUs (include/breakid_multi.h)
  bool fast = false, filter = true;
  int opt, li;
  optind = 0;
  while ((opt = getopt_long_only(argc, argv, "h?", longopts, &li)) != -1)
  {
    switch (opt)
    {
    case 'h': case '?': std::cerr << HELP; exit(1);
    case 1: inp_file = optarg; break;
    case 2: out_file = optarg; break;
    case 3: qual = (int) std::labs(atol(optarg)); break;
    case 4: nib_dir = optarg; break;
    case 5: fast = true; break;
    case 6: break;  // the reference dereferences a NULL optarg here (has_arg = 0); `times` is effectively always 2
    case 7: filter = false; break;
    case 8: device = atoi(optarg); break;
    case 9: n_gpus = atoi(optarg); break;
    case 10: transport = !strcmp(optarg, "rccl") ? BK_TRANSPORT_RCCL : !strcmp(optarg, "local") ? BK_TRANSPORT_LOCAL : BK_TRANSPORT_AUTO; break;
    default: std::cerr << "Error: cannot parse arguments.\n"; exit(1);
    }
  }
  if (inp_file.empty() || out_file.empty())
  {
    std::cerr << HELP << "Error: input- and output file is required.\n";
    exit(1);
  }
  if (nib_dir.empty())
  {
    std::cerr << HELP << "Error: nib file's root dir is required.\n";
    exit(1);
  }
  std::cout << "start to stats the insert size...\n";
  // feed: the GPU decoder first (BGZF inflate + record decode on the device; every htslib-written BAM qualifies), the
  // host decoder (all cores, pinned columns) for files whose records straddle BGZF blocks or that exceed one batch.
  // BREAKID_HOST_DECODE=1 forces the host path.
  char err[512];
  bk_bam *bam = nullptr;
  bk_bam_dev *dbam = nullptr;
  int nt = 0;
  const char *const *names = nullptr;
  const uint32_t *lens = nullptr;
  bk_soa soa;
  int soa_where = BK_MEM_HOST;
  {
    FILE *probe = fopen(inp_file.c_str(), "rb");
    if (!probe)
    {
      std::cerr << "Error: can not open bam-file: " << inp_file << std::endl;
      exit(1);
    }
    fclose(probe);
  }
  const bool multi = n_gpus >= 1;  // the sharded run: every rank decodes its part of the file on its own GPU (below), or takes its range of the host table
  bk_ctx *ctx = nullptr;
  auto host_decode = [&] {
    dbam = nullptr;
    if (bk_bam_open(inp_file.c_str(), &bam, err, sizeof err) != BK_OK)
    {
      std::cerr << "Error: can not open bam-file: " << inp_file << std::endl;
      exit(1);
    }
    bk_bam_header(bam, &nt, &names, &lens);
    if (bk_bam_decode(bam, &soa, err, sizeof err) != BK_OK)
    {
      std::cerr << "Error: " << err << std::endl;
      exit(1);
    }
  };
  const bool multi_from_file = multi && !getenv("BREAKID_HOST_DECODE");
  // one read of the file: BGZF inflate + record decode on the device, the stream pass of the hot path running on the chunks
  // already decoded while the rest of the file is still arriving (the reference reads the BAM twice, BreakID.cc:1929, :1414)
  if (!multi && !getenv("BREAKID_HOST_DECODE") && bk_bam_decode_device_ctx(inp_file.c_str(), device, qual, &dbam, &ctx, &nt, &names, &lens, err, sizeof err) == BK_OK)
  {
    soa_where = BK_MEM_DEVICE;
    bk_feed_release_caches();  // this process decodes one file: the feed's staging buffers and slots (1-2.5 GB of device memory) go back
  }
  else if (!multi_from_file)
    host_decode();
  {
    std::ifstream rn((nib_dir + "/ref_names.txt").c_str());
    if (!rn.is_open())
    {
      std::cerr << "Error: cannot open reference names file.\n";
      exit(1);
    }
  }
  auto die = [&](int rc) {
    std::cerr << (rc == BK_ERR_CIGAR ? "error cigar: " : bk_last_error(ctx)) << std::endl;
    exit(rc == BK_ERR_CIGAR ? -1 : 1);
  };
  int rc;
  double w = 0;
  clock_t scan_start = clock(), scan_end = scan_start, cluster_start = scan_start, cluster_end = scan_start, bp_start = scan_start, bp_end = scan_start;
  uint64_t n_pairs = 0, n_clustered = 0, n_valid = 0, n_clusters = 0;
  uint32_t n_groups = 0;
  auto need_index = [&] {  // findEncompassingReadsAndBreakPointInfo loads the index for every group that reaches it (:405-416)
    if (!index_loads(inp_file))
    {
      std::cerr << "Error: please index bam-file first:\t" << inp_file << std::endl;
      exit(1);
    }
  };
  if (multi)
  {
    // one sample over n_gpus GPUs: record ranges per rank, RCCL (or in-process) exchange of the small tables
    // the GPU feed per rank first (bk_bam_decode_device_part); files it cannot cut into parts (records across BGZF blocks) and
    // anything else it refuses go through the host decoder and the record ranges of its table
    rc = BK_ERR_IO;
    if (multi_from_file) rc = bk_multi_run_bam(inp_file.c_str(), n_gpus, transport, qual, fast ? 1 : 0, &w, &n_clustered, &ctx, &nt, &names, &lens, err, sizeof err);
    if (rc != BK_OK && (!multi_from_file || rc == BK_ERR_IO || rc == BK_ERR_LIMIT))
    {
      host_decode();
      rc = bk_multi_run(&soa, lens, names, nt, n_gpus, transport, qual, fast ? 1 : 0, &w, &n_clustered, &ctx, err, sizeof err);
    }
    if (rc != BK_OK)
    {
      std::cerr << (rc == BK_ERR_CIGAR ? "error cigar: " : err) << std::endl;
      exit(rc == BK_ERR_CIGAR ? -1 : 1);
    }
    double mean = 0, sd = 0;
    (void) bk_multi_stats(ctx, &mean, &sd, nullptr, nullptr);
    std::cout << "the insert size mean: " << mean << ", the insert size sd:" << sd << " .\n";
    std::cout << "cluster_dist = span_dist = mask_dist = scan_dist = " << w << " .\n";
    std::cout << "Scanning discordant read pairs ...\n";
    std::cout << "Scanning discordant read pairs done.\n";
    if (n_clustered) need_index();
  }
  else
  {
    if (!ctx)  // host decoder: the table is uploaded now (the GPU feed has attached it and run the stream pass already)
    {
      if (bk_init(device, lens, names, nt, &ctx) != BK_OK)
      {
        std::cerr << "Error: " << bk_last_error(nullptr) << std::endl;
        exit(1);
      }
      if ((rc = bk_upload_records(ctx, &soa, soa_where)) != BK_OK) die(rc);
    }
    double mean = 0, sd = 0;
    if ((rc = bk_isize_stats(ctx, &mean, &sd)) != BK_OK) die(rc);
    std::cout << "the insert size mean: " << mean << ", the insert size sd:" << sd << " .\n";
    const int times = 2;
    w = times * std::sqrt(times) * (mean + 3 * sd);
    std::cout << "cluster_dist = span_dist = mask_dist = scan_dist = " << w << " .\n";
    scan_start = clock();
    std::cout << "Scanning discordant read pairs ...\n";
    if ((rc = bk_discordant_pairs(ctx, qual, w, &n_pairs, &n_groups)) != BK_OK) die(rc);
    std::cout << "Scanning discordant read pairs done.\n";
    scan_end = clock();
    cluster_start = clock();
    if ((rc = bk_mask_and_cluster(ctx, w, fast ? 1 : 0, &n_clustered)) != BK_OK) die(rc);
    cluster_end = clock();
    bp_start = clock();
    if ((rc = bk_split_evidence(ctx, nullptr)) != BK_OK) die(rc);
    if ((rc = bk_cluster_summary(ctx, w, &n_clusters)) != BK_OK) die(rc);
    if (n_clustered) need_index();
    if ((rc = bk_split_breakpoints(ctx, w, &n_valid)) != BK_OK) die(rc);
    bp_end = clock();
  }
  const void *data = nullptr;
  uint64_t cnt = 0;
  if ((rc = bk_fetch(ctx, BK_STAGE_CLUSTERS, &data, &cnt, nullptr, nullptr)) != BK_OK) die(rc);
  const bk_cluster *cl = (const bk_cluster *) data;
  if (multi)
  {
    n_valid = 0;
    for (uint64_t i = 0; i < cnt; ++i) n_valid += (cl[i].flags & 2u) != 0;
  }
  std::cout << "valid cluster count: " << n_valid << std::endl;
  // annotate_cluster_for_sa_tag (BreakID.cc:492-567)
  vector<OutRow> rows;
  vector<Txpt> txpts;
  bool have_valid = false;
  for (uint64_t i = 0; i < cnt; ++i) have_valid |= (cl[i].flags & 2u) != 0;
  if (n_clustered >= 1 || have_valid)
  {
    // the reference reads refGene.txt for every group that reaches findClusterBreakPointInfoSaTag and exits if it is missing
    const char *inst = getenv("BREAKID_INSTALLDIR");
    string ref_gene = string(inst ? inst : BREAKID_INSTALLDIR) + "/ref_files/refGene.txt";
    if (!read_refgene(ref_gene, txpts))
    {
      std::cerr << "Error: cannot open \t" << ref_gene << std::endl;
      exit(1);
    }
  }
  for (uint64_t i = 0; i < cnt; ++i)
  {
    if (!(cl[i].flags & 2u)) continue;
    OutRow r;
    r.c = cl[i];
    r.p1_chr = cl[i].p1_tid < 0 ? "*" : names[cl[i].p1_tid];
    r.p2_chr = cl[i].p2_tid < 0 ? "*" : names[cl[i].p2_tid];
    long p1 = (long) cl[i].p1_exact, p2 = (long) cl[i].p2_exact;  // exact positions are never -1 for valid clusters
    annotate_side(txpts, r.p1_chr, p1, r.g1, r.e1, r.s1);
    annotate_side(txpts, r.p2_chr, p2, r.g2, r.e2, r.s2);
    r.rpt1 = neighbour_seq(nib_dir, r.p1_chr, (int32_t) cl[i].p1_exact);
    r.rpt2 = neighbour_seq(nib_dir, r.p2_chr, cl[i].p2_exact);
    r.is_rpt = longest_run(r.rpt1) > 10 || longest_run(r.rpt2) > 10;
    r.af1 = (float) (long) cl[i].n_sr / (float) (double) cl[i].depth1;  // :475-478
    r.af2 = (float) (long) cl[i].n_sr / (float) (double) cl[i].depth2;
    rows.push_back(r);
  }
  // write_enspan_out (BreakID.cc:1184-1263): std::sort with the reference's comparator
  std::sort(rows.begin(), rows.end(), cmp_cluster);
  std::ofstream out, outf;
  if (!filter)
  {
    out.open((out_file + "_fusion_all.txt").c_str());
    out << HEADER;
  }
  outf.open((out_file + "_fusion.txt").c_str());
  outf << HEADER;
  for (auto &r : rows)
  {
    bool all_ok = r.c.n_sr > 0 && r.c.p1_exact != 0xFFFFFFFFu && r.c.p2_exact != -1;
    bool filt_ok = all_ok && (!(r.g1 == "intergenic" && r.g2 == "intergenic") && r.g1 != r.g2) && !r.is_rpt;
    if (filt_ok) write_row(outf, r);
    if (!filter && all_ok) write_row(out, r);
  }
  if (!filter) out.close();
  outf.close();
  {
    std::ofstream p((out_file + "_params.txt").c_str());  // write_enspan_params :1170-1182
    p << "ENSPAN" << std::endl;
    p << "inp_file\t" << inp_file << std::endl;
    p << "out_file\t" << out_file << std::endl;
    p << "qual\t" << (long) qual << std::endl;
    p << "w\t" << w << std::endl;
    p << "build\t" << build << std::endl;
  }
  clock_t end = clock();
  std::cout << "the fusion process of file " << inp_file << "  costs time: " << (end - start) / double(CLOCKS_PER_SEC) << " seconds" << std::endl;
  {
    // :175-191.  scan_pairs_count and after_cluster_count are never updated by the reference (always 0);
    // removed_isolated_pair_count sums the groups that keep >= 2 pairs (:128); root_cluster_num is what the clustering of the
    // LAST such group returned (:131-136; the reference leaves it uninitialised when no group qualifies - 0 here)
    const bk_group_stat *gs = nullptr;
    uint32_t ngs = 0;
    if (!multi && (rc = bk_group_stats(ctx, &gs, &ngs)) != BK_OK) die(rc);
    if (multi && (rc = bk_multi_stats(ctx, nullptr, nullptr, &gs, &ngs)) != BK_OK) die(rc);  // (summed over the ranks)
    int removed_isolated_pair_count = 0, root_cluster_num = 0;
    for (uint32_t g = 0; g < ngs; ++g)
      if (gs[g].n_isolated_removed >= 2)
      {
        removed_isolated_pair_count += (int) gs[g].n_isolated_removed;
        root_cluster_num = fast ? (gs[g].cluster_id_end ? (int) gs[g].cluster_id_end - 1 : 0)
                                : (int) (gs[g].cluster_id_end + gs[g].n_isolated_removed - gs[g].n_clustered);
      }
    std::ofstream p((out_file + "_performance.txt").c_str());
    p << "scan_dist\tdiscordant pairs\tremove isolated\tafter_cluster\troot cluster\tscanning time\tcluster time\tfind breakpoint time\ttotal time" << std::endl;
    p << w << "\t" << 0 << "\t" << removed_isolated_pair_count << "\t" << 0 << "\t" << root_cluster_num << "\t" << (scan_end - scan_start) / double(CLOCKS_PER_SEC)
      << "\t" << (cluster_end - cluster_start) / double(CLOCKS_PER_SEC) << "\t" << (bp_end - bp_start) / double(CLOCKS_PER_SEC) << "\t"
      << (end - start) / double(CLOCKS_PER_SEC) << std::endl;
  }
  if (multi)
    bk_multi_free(ctx);
  else
    bk_free(ctx);
  if (bam) bk_bam_close(bam);
  if (dbam) bk_bam_dev_free(dbam);
  return 0;
}
