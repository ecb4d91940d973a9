// The compiled `rela` module on libhsad.so: same names and constructor signatures as the reference's rela/pybind.cc:16-93.
#include "hsad_host.h"

namespace py = pybind11;
using namespace hsadpy;

PYBIND11_MODULE(rela, m) {
  m.doc() = "rela (rela/pybind.cc) on the MI355X device pipeline: libhsad.so behind the reference's class names";
  // the batch type: seven attributes holding torch tensors (or dicts of them), constructible from Python like the mirror's
  {
    py::class_<RNNTransition, std::shared_ptr<RNNTransition>> t(m, "RNNTransition");
    t.def(py::init([](py::object obs, py::object action, py::object reward, py::object terminal, py::object bootstrap, py::object seq_len) {
      return std::make_shared<RNNTransition>(RNNTransition{obs, py::dict(), action, reward, terminal, bootstrap, seq_len});
    }));
#define HSAD_FIELD(name) t.def_readwrite(#name, &RNNTransition::name)
    HSAD_FIELD(obs);
    HSAD_FIELD(h0);
    HSAD_FIELD(action);
    HSAD_FIELD(reward);
    HSAD_FIELD(terminal);
    HSAD_FIELD(bootstrap);
    HSAD_FIELD(seq_len);
#undef HSAD_FIELD
  }

  // capacity, seed, alpha (priority exponent), beta (importance exponent), prefetch (accepted, unused: sampling is a stream-ordered kernel)
  {
    py::class_<RNNPrioritizedReplay, std::shared_ptr<RNNPrioritizedReplay>> r(m, "RNNPrioritizedReplay");
    r.def(py::init<int, int, float, float, int>(), py::arg("capacity"), py::arg("seed"), py::arg("alpha"), py::arg("beta"), py::arg("prefetch"));
    r.def("size", &RNNPrioritizedReplay::size, "sequences stored (synchronises the producers' stream)");
    r.def("num_add", &RNNPrioritizedReplay::num_add, "sequences added so far");
    r.def("sample", &RNNPrioritizedReplay::sample, py::arg("batchsize"), py::arg("device"), "-> (RNNTransition of [T, B, ...] tensors, weight [B])");
    r.def("update_priority", &RNNPrioritizedReplay::update_priority, py::arg("priority"));
  }

  py::class_<ThreadLoop, std::shared_ptr<ThreadLoop>>(m, "ThreadLoop");

  {
    py::class_<Context> c(m, "Context");
    c.def(py::init<>());
    c.def("push_env_thread", &Context::push_env_thread, py::keep_alive<1, 2>(), py::arg("loop"), "-> number of loops attached");
    for (auto&& [name, fn] : {std::pair<const char*, void (Context::*)()>{"start", &Context::start}, {"pause", &Context::pause}, {"resume", &Context::resume},
                              {"terminate", &Context::terminate}})
      c.def(name, fn);
    c.def("terminated", &Context::terminated, "every attached (evaluation) loop has finished, or terminate() was called");
  }

  {
    py::class_<R2D2Actor, std::shared_ptr<R2D2Actor>> a(m, "R2D2Actor");
    // training: (runner, multi_step, batchsize, gamma, eta, seq_len, num_player, replay); evaluation: (runner, num_player)
    a.def(py::init<std::shared_ptr<BatchRunner>, int, int, float, float, int, int, std::shared_ptr<RNNPrioritizedReplay>>());
    a.def(py::init<std::shared_ptr<BatchRunner>, int>());
    a.def("num_act", &R2D2Actor::num_act);
  }

  {
    py::class_<BatchRunner, std::shared_ptr<BatchRunner>> b(m, "BatchRunner");
    b.def(py::init<py::object, const std::string&, int, const std::vector<std::string>&>(), py::arg("agent"), py::arg("device"), py::arg("max_batchsize"),
          py::arg("methods"));
    b.def("start", &BatchRunner::start).def("stop", &BatchRunner::stop);
    b.def("update_model", &BatchRunner::update_model, py::arg("agent"), "copy agent.state_dict() into the acting nets (BatchRunner::updateModel)");
  }

  m.def("aggregate_priority", &aggregate_priority, py::arg("priority"), py::arg("seq_len"), py::arg("eta"));
}
