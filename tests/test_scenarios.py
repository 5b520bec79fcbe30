"""The reference's integration-test scenarios at the hot-path boundary: oracle on CPU, engine on GPU."""
import pytest

import backends
import scenarios


@pytest.mark.parametrize("scenario", scenarios.ALL, ids=lambda f: f.__name__)
def test_oracle(scenario):
    scenario(backends.OracleLeader)


@pytest.mark.gpu
@pytest.mark.parametrize("scenario", scenarios.ALL, ids=lambda f: f.__name__)
def test_engine(scenario, rg):
    scenario(backends.EngineLeader)


@pytest.mark.parametrize("scenario", scenarios.FLOW, ids=lambda f: f.__name__)
def test_oracle_flow_control(scenario):
    scenario(backends.OracleLeader)


@pytest.mark.gpu
@pytest.mark.parametrize("scenario", scenarios.FLOW, ids=lambda f: f.__name__)
def test_engine_flow_control(scenario, rg):
    scenario(backends.EngineLeader)
