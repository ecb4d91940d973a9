// The compiled `rela` module on libhsad.so: same names and constructor signatures as the reference's rela/pybind.cc:16-93.
#include "hsad_host.h"

namespace py = pybind11;
using namespace hsadpy;

PYBIND11_MODULE(rela, m) {
  m.doc() = "rela (rela/pybind.cc) on the MI355X device pipeline: libhsad.so behind the reference's class names";
  py::class_<RNNTransition, std::shared_ptr<RNNTransition>>(m, "RNNTransition")
      .def(py::init([](py::object obs, py::object action, py::object reward, py::object terminal, py::object bootstrap, py::object seq_len) {
             return std::make_shared<RNNTransition>(RNNTransition{obs, py::dict(), action, reward, terminal, bootstrap, seq_len});
           }))
      .def_readwrite("obs", &RNNTransition::obs)
      .def_readwrite("h0", &RNNTransition::h0)
      .def_readwrite("action", &RNNTransition::action)
      .def_readwrite("reward", &RNNTransition::reward)
      .def_readwrite("terminal", &RNNTransition::terminal)
      .def_readwrite("bootstrap", &RNNTransition::bootstrap)
      .def_readwrite("seq_len", &RNNTransition::seq_len);

  py::class_<RNNPrioritizedReplay, std::shared_ptr<RNNPrioritizedReplay>>(m, "RNNPrioritizedReplay")
      .def(py::init<int, int, float, float, int>())      // capacity, seed, alpha, beta, prefetch
      .def("size", &RNNPrioritizedReplay::size)
      .def("num_add", &RNNPrioritizedReplay::num_add)
      .def("sample", &RNNPrioritizedReplay::sample)
      .def("update_priority", &RNNPrioritizedReplay::update_priority);

  py::class_<ThreadLoop, std::shared_ptr<ThreadLoop>>(m, "ThreadLoop");

  py::class_<Context>(m, "Context")
      .def(py::init<>())
      .def("push_env_thread", &Context::push_env_thread, py::keep_alive<1, 2>())
      .def("start", &Context::start)
      .def("pause", &Context::pause)
      .def("resume", &Context::resume)
      .def("terminate", &Context::terminate)
      .def("terminated", &Context::terminated);

  py::class_<R2D2Actor, std::shared_ptr<R2D2Actor>>(m, "R2D2Actor")
      .def(py::init<std::shared_ptr<BatchRunner>, int, int, float, float, int, int, std::shared_ptr<RNNPrioritizedReplay>>())
      .def(py::init<std::shared_ptr<BatchRunner>, int>())      // evaluation mode
      .def("num_act", &R2D2Actor::num_act);

  py::class_<BatchRunner, std::shared_ptr<BatchRunner>>(m, "BatchRunner")
      .def(py::init<py::object, const std::string&, int, const std::vector<std::string>&>())
      .def("start", &BatchRunner::start)
      .def("stop", &BatchRunner::stop)
      .def("update_model", &BatchRunner::update_model);

  m.def("aggregate_priority", &aggregate_priority);
}
