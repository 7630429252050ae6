from blades_amd.parallel.communicator import Communicator, CommunicationSet, _BufferList  # noqa: F401
