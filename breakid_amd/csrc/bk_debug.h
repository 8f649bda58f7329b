#pragma once
#include <cstdlib>
#include <string>

// Debugging aids are asked for by name in ONE variable: BK_DEBUG=lanes,svc,sort (comma-separated; INTEGRATION.md section 5 lists them).
inline bool bk_debug(const char *what)
{
  const char *e = getenv("BK_DEBUG");  // (read at every call: most callers keep the answer in a static)
  if (!e || !*e) return false;
  const std::string all = std::string(",") + e + ",";
  return all.find(std::string(",") + what + ",") != std::string::npos;
}
