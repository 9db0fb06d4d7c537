// DPGO_solver.h -- included by src/PGOAgentROS.cpp:10; the solver types live in DPGO_types.h and the
// local solve itself runs on the GPU behind dpgo_agent_iterate().
#pragma once
#include "DPGO_types.h"
