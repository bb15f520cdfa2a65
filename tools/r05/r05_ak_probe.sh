#!/bin/bash
# round 5, step ak: timing-only probe of DESIGN section 10's first item -- is the END of the publish-only tail kernel on a chained step's critical
# path?  tools/variants/linger200.so: the scalars' workgroup of a publish-only tail lingers 2 us AFTER everything is published (the host gets the
# result when it always did; the kernel ends 2 us later); linger0.so: the same build without the wait.  (Variants of solve_kernels.hip with a
# temporary -DMPPI_DIAG_TAIL_LINGER block behind publish_entry of the trajectory cost.)
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/abn.sh r05_ak_cfg3 3 "tools/variants/linger0.so tools/variants/linger200.so"
