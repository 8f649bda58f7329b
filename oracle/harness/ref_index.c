/* TEST INFRASTRUCTURE ONLY: build <bam>.bai with the reference's vendored samtools-1.3.1/htslib. */
#include <stdio.h>
#include "bam.h"
int main(int argc, char **argv)
{
  if (argc != 2) { fprintf(stderr, "usage: ref_index in.bam\n"); return 2; }
  return bam_index_build(argv[1]) == 0 ? 0 : 1;
}
