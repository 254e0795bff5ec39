"""CPU oracle (test infrastructure only): see nl_oracle.h."""
