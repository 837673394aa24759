// Drop-in include path: a caller's  #include "kmerhash/hashmap_robinhood.hpp"  resolves here when
// -I<kmerhash_amd>/include precedes the reference's include directory.  Declares
// fsc::hashmap_robinhood_doubling backed by libkmerhash_amd.so (see kmerhash_amd/hashmap.hpp).
#ifndef KMERHASH_AMD_DROPIN_HASHMAP_ROBINHOOD_HPP_
#define KMERHASH_AMD_DROPIN_HASHMAP_ROBINHOOD_HPP_
#include "../kmerhash_amd/hashmap.hpp"
#endif
