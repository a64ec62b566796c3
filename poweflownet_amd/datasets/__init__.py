from .PowerFlowData import PowerFlowData, denormalize, random_bus_type  # noqa: F401
