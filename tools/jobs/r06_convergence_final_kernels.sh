# 300 epochs of IQL + SAD (configs[2]'s learner) on the final round-6 recurrence launches (running counters, straight-line tile product):
# compared with the first 300 epochs of the 1,200-epoch runs of the same round (seeds 11 / 23)
set -u
O=gpurun_out; mkdir -p $O
rm -rf /tmp/c_fin
timeout ${TMO:-1700} python -m hanabi_sad_amd.selfplay --sad 1 --num_game 6400 --num_thread 80 --num_game_per_thread 80 --batchsize 128 \
  --replay_buffer_size 131072 --burn_in_frames 10000 --num_epoch ${EPOCHS:-300} --epoch_len 1000 --num_eval_game 1000 --seed ${SEED:-11} --save_dir /tmp/c_fin > $O/r06_conv_final.out 2>&1
cp /tmp/c_fin/train.log $O/r06_selfplay_convergence_final_kernels_train.log 2>/dev/null; grep "eval score" $O/r06_conv_final.out | tail -3
