// TEST INFRASTRUCTURE ONLY.  Builds the real reference executable from where its sources lie.
// The reference bakes its install directory into src/installdir.h (a path that does not exist in
// this image; upstream regenerates it with generate_installDIR.sh).  Here the macro is re-pointed
// at $BREAKID_REF_INSTALLDIR at run time so that <dir>/ref_files/refGene.txt can be a synthesised
// fixture (the real refGene.txt is a missing large blob, /root/reference/.MISSING_LARGE_BLOBS).
#include <cstdlib>
#include "/root/reference/src/BreakID.h"
static inline const char *oracle_installdir()
{
  const char *e = getenv("BREAKID_REF_INSTALLDIR");
  return e ? e : "/nonexistent";
}
#undef INSTALLDIR
#define INSTALLDIR oracle_installdir()
#include "/root/reference/src/BreakID.cc"
