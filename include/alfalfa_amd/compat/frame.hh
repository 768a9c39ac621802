// Drop-in for the reference's "frame.hh": see decoder.hh in this directory.  Everything is in alfalfa.hh.
#pragma once
#ifndef ALFALFA_AMD_GLOBAL_NAMES
#define ALFALFA_AMD_GLOBAL_NAMES
#endif
#include "../alfalfa.hh"
