// Empty stand-in (written here, TEST INFRASTRUCTURE): the reference's rasterizer.h includes the CUDA context header for its GPU
// half, which the CPU half compiled into oracle/_ref never touches.  See oracle/ref_build.py.
#pragma once
