// Drop-in for the reference's "ivf_reader.hh": code written against excamera/alfalfa's headers compiles against the MI355X
// implementation by putting this directory first on the include path.  Everything is in alfalfa.hh.
#pragma once
#ifndef ALFALFA_AMD_GLOBAL_NAMES
#define ALFALFA_AMD_GLOBAL_NAMES
#endif
#include "../alfalfa.hh"
