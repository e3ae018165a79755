cd $GRAFT_REPO_ROOT
timeout 600 python -m challenge_amd.sj_train --synthetic --epochs 1 --steps_per_epoch 10 --validation_steps 2 --batch_size 16 --n_frame 128 --v 9 --name smoke 2>&1 | tail -6
rm -f *.pt *.csv
