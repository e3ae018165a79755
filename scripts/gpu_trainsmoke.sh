# 1-epoch synthetic training smoke on the GPU box; artefacts go to a temp directory, never the repo root
cd $GRAFT_REPO_ROOT
OUT=$(mktemp -d)
(cd $OUT && PYTHONPATH=$GRAFT_REPO_ROOT timeout 600 python -m challenge_amd.sj_train --synthetic --epochs 1 --steps_per_epoch 10 --validation_steps 2 --batch_size 16 --n_frame 128 --v 9 --name smoke 2>&1 | tail -6; ls $OUT)
rm -rf $OUT
