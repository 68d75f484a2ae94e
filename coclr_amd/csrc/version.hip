#include "../../include/coclr_hip.h"
extern "C" int coclr_abi_version(void) { return 20; }
