#!/bin/bash
echo "== victim: ONLY the +-i rotations packed (crossed op_sel), the rest scalar"; timeout 300 build/repro_rot 20000 17 2>&1
echo "== victim: every packed form of the product EXCEPT the crossed ones (the shipped helpers: DSA_PK_CROSSED=0)"; timeout 300 build/repro_rot_plain 20000 0 2>&1
echo "== victim: all packed forms incl. the crossed ones (the helpers of rounds 2-5)"; timeout 300 build/repro_rot_allcrossed 20000 17 2>&1
