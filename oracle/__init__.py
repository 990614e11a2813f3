"""CPU oracle (test infrastructure).  See oracle/shader_oracle.cpp."""
