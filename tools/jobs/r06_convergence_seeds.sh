# two more 1,200-epoch convergence runs on the final loop: seeds 11 and 23 next to the seed-7 runs of rounds 2-5;
# the configuration of profiles/r05_selfplay_convergence_1200_train.log.  ~35 min each.
set -u
O=gpurun_out; mkdir -p $O
for seed in ${SEEDS:-11 23}; do
  rm -rf /tmp/conv_$seed
  timeout ${PER_RUN_TIMEOUT:-2700} python -m hanabi_sad_amd.selfplay --sad 1 --num_game 6400 --num_thread 80 --num_game_per_thread 80 --batchsize 128 \
    --replay_buffer_size 131072 --burn_in_frames 10000 --num_epoch ${EPOCHS:-1200} --epoch_len 1000 --num_eval_game 1000 --seed $seed \
    --save_dir /tmp/conv_$seed > $O/r06_conv_seed${seed}.out 2>&1
  cp /tmp/conv_$seed/train.log $O/r06_selfplay_convergence_1200_seed${seed}_train.log 2>/dev/null
  grep "eval score" $O/r06_conv_seed${seed}.out | tail -3
done
