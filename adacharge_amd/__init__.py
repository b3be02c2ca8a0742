"""adacharge_amd: MI355X-native batched MPC solver behind adacharge's API."""
