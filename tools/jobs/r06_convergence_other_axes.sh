# the other method / auxiliary-task axes on the round-6 code, 300 epochs each like the round-2 runs they are compared with:
#  aux: IQL + --pred_weight 0.25 (configs[3]'s learner; B = 128 -> the 16-row x 64-unit BPTT launch), seed 11
#  vdn: --method vdn --shuffle_color 1 --sad 0 (the Other-Play setting; 256 agent rows per batch -> the 32 x 32 BPTT launch), seed 3
set -u
O=gpurun_out; mkdir -p $O
rm -rf /tmp/c_aux /tmp/c_vdn
timeout 1500 python -m hanabi_sad_amd.selfplay --sad 1 --pred_weight 0.25 --num_game 6400 --num_thread 80 --num_game_per_thread 80 --batchsize 128 \
  --replay_buffer_size 131072 --burn_in_frames 10000 --num_epoch 300 --epoch_len 1000 --num_eval_game 1000 --seed 11 --save_dir /tmp/c_aux > $O/r06_conv_aux.out 2>&1
cp /tmp/c_aux/train.log $O/r06_selfplay_convergence_aux_train.log 2>/dev/null; grep "eval score" $O/r06_conv_aux.out | tail -2
timeout 1500 python -m hanabi_sad_amd.selfplay --method vdn --shuffle_color 1 --sad 0 --num_game 6400 --num_thread 80 --num_game_per_thread 80 --batchsize 128 \
  --replay_buffer_size 131072 --burn_in_frames 10000 --num_epoch 300 --epoch_len 1000 --num_eval_game 1000 --seed 3 --save_dir /tmp/c_vdn > $O/r06_conv_vdn.out 2>&1
cp /tmp/c_vdn/train.log $O/r06_selfplay_convergence_op_vdn_train.log 2>/dev/null; grep "eval score" $O/r06_conv_vdn.out | tail -2
