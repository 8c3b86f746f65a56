// vq_group.h -- the in-process group as the other host files see it (internal; the exported API is e2vq_group_* in
// include/ecoz2_vq.h): an opaque object, the session hook of a rank, failure propagation.
#pragma once
#include "vq_session.h"

#include <string>

struct E2Group;  // (vq_group.cpp)
// devices[r] = HIP device of rank r.  collective: "rccl", "p2p" or "" (RCCL when every rank has a device of its own and
// librccl.so loads, else the peer-to-peer kernel).  Returns null with the error message set.
E2Group* e2g_create(int world, const int* devices, const std::string& coll, bool verbose);
void e2g_destroy(E2Group* g);
// the exchange of rank r as a session hook (e2vq_set_allreduce); *force: call it even in a group of one
void e2g_hook(E2Group* g, int r, e2vq_allreduce_fn* fn, void** user, bool* force);
void e2g_fail(E2Group* g);                  // a rank gives up: the others leave their rendezvous with an error
const volatile bool* e2g_failed_flag(E2Group* g);
std::string e2g_first_error(E2Group* g);    // message of the rank that failed first
bool e2g_uses_rccl(E2Group* g);
int e2g_world(E2Group* g);
int e2g_device(E2Group* g, int r);
const char* e2g_what(E2Group* g);           // one line describing the exchange
void e2g_rccl_traffic(E2Group* g, int r, long* calls, long* bytes);
