// edge_kernel.hpp -- K3 helpers for batched swept-volume edge validation (filled in below).
#pragma once
